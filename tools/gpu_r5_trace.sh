#!/bin/bash
# GPU box: kernel trace of the one-shot probe at the batch sizes in $2 (default 128,256), by-grid summary.  usage: tools/gpu_r5_trace.sh TAG [BATCHES]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=${1:-r5trace}; sizes=${2:-128,256}
for B in ${sizes//,/ }; do
  mkdir -p gpurun_out/${tag}_prof$B
  PTTS_PROBE_BATCHES=$B PTTS_PROBE_REPS=1 timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_prof$B -o prof -- python3 tools/big_batch_probe.py > gpurun_out/${tag}_prof$B.log 2>&1 || { echo "trace B=$B failed"; tail -5 gpurun_out/${tag}_prof$B.log; exit 1; }
  python3 tools/trace_summary.py $(ls gpurun_out/${tag}_prof$B/*kernel_trace.csv | head -1) 60 > gpurun_out/${tag}_by_grid_b$B.txt
  rm -rf gpurun_out/${tag}_prof$B
  echo "== B=$B"; head -42 gpurun_out/${tag}_by_grid_b$B.txt
done
