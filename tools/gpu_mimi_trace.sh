#!/bin/bash
# GPU box: per-kernel times of batch-64 x 125-frame Mimi decodes (tools/pmc_mimi.py, PTTS_PMC_REPS decodes) + optionally the SQ counters of one kernel.
# usage: tools/gpu_mimi_trace.sh TAG [KERNEL_SUBSTRING_FOR_SQ]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/${1}_tr
PTTS_PMC_REPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${1}_tr -o tr -- python3 tools/pmc_mimi.py > gpurun_out/${1}_tr.log 2>&1 || { tail -3 gpurun_out/${1}_tr.log; exit 1; }
python3 tools/trace_summary.py $(ls gpurun_out/${1}_tr/*kernel_trace.csv | head -1) 40 > gpurun_out/${1}_mimi_by_grid.txt; cat gpurun_out/${1}_mimi_by_grid.txt
rm -f gpurun_out/${1}_tr/*kernel_trace.csv
if [ -n "$2" ]; then
  mkdir -p gpurun_out/${1}_sq
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d gpurun_out/${1}_sq -o pmc -- python3 tools/pmc_mimi.py > gpurun_out/${1}_sq.log 2>&1 || { tail -3 gpurun_out/${1}_sq.log; exit 1; }
  python3 tools/pmc_summary.py $(ls gpurun_out/${1}_sq/*counter_collection.csv | head -1) "$2" | tee gpurun_out/${1}_sq.txt
  rm -f gpurun_out/${1}_sq/*kernel_trace.csv
fi
