#!/bin/bash
# GPU box: LDS / SQ counters of one kernel in a batch-64 Mimi decode.  usage: tools/gpu_pmc_one.sh TAG KERNEL_SUBSTRING
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/${1}_pm
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/${1}_pm -o pmc -- python3 tools/pmc_mimi.py > gpurun_out/${1}_pm.log 2>&1 || { tail -3 gpurun_out/${1}_pm.log; exit 1; }
python3 tools/pmc_summary.py $(ls gpurun_out/${1}_pm/*counter_collection.csv | head -1) "$2"
rm -f gpurun_out/${1}_pm/*kernel_trace.csv
