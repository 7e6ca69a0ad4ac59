#!/bin/bash
# GPU box: round 5's evidence from the build in the tree.  usage: tools/gpu_r5_final.sh PART   (PART = a: tests + bench + kernel stats, b: counters + traces)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=r5fin
if [ "$1" = a ]; then
  tools/gpu_tests.sh $tag || exit 1
  timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
  tail -c 400 gpurun_out/${tag}_bench.json; echo
  mkdir -p gpurun_out/${tag}_prof
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o prof -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-two-engines --no-traffic > gpurun_out/${tag}_prof.log 2>&1; echo "rocprof rc=$?"
  python3 tools/trace_summary.py $(ls gpurun_out/${tag}_prof/*kernel_trace.csv | head -1) 70 > gpurun_out/${tag}_by_grid.txt; head -20 gpurun_out/${tag}_by_grid.txt
  cp $(ls gpurun_out/${tag}_prof/*kernel_stats.csv | head -1) gpurun_out/${tag}_kernel_stats.csv
  rm -f gpurun_out/${tag}_prof/*kernel_trace.csv
else
  mkdir -p gpurun_out/${tag}_sq
  PTTS_PROBE_STEPS=12 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${tag}_sq -o sq -- python3 tools/traffic_probe.py > gpurun_out/${tag}_sq.log 2>&1; echo "sq rc=$?"
  python3 tools/pmc_summary.py $(ls gpurun_out/${tag}_sq/*counter_collection.csv | head -1) k_skinny > gpurun_out/${tag}_pmc_ar_sq.txt
  python3 tools/pmc_summary.py $(ls gpurun_out/${tag}_sq/*counter_collection.csv | head -1) k_attn_step >> gpurun_out/${tag}_pmc_ar_sq.txt
  rm -f gpurun_out/${tag}_sq/*kernel_trace.csv
  head -8 gpurun_out/${tag}_pmc_ar_sq.txt | cut -c1-200
  bash tools/gpu_mimi_trace.sh ${tag} > gpurun_out/${tag}_mimi_trace.log 2>&1; echo "mimi trace rc=$?"
  bash tools/gpu_pmc_mimi.sh ${tag}m > gpurun_out/${tag}_pmc_mimi.log 2>&1; echo "pmc mimi rc=$?"
  bash tools/gpu_r5_trace.sh ${tag} 128,256 > gpurun_out/${tag}_wide_trace.log 2>&1; echo "wide trace rc=$?"
fi
