#!/bin/bash
# GPU box: per-kernel times of the batch-64 Mimi decode under environment settings.  usage: tools/gpu_mimi_ab.sh TAG KERNEL "ENV1" "ENV2" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1; kern=$2; shift 2
i=0
for e in "$@"; do
  i=$((i+1)); d=gpurun_out/${tag}_ab$i; mkdir -p $d
  env $e PTTS_PMC_REPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -o tr -- python3 tools/pmc_mimi.py > $d.log 2>&1 || { echo "$e failed"; tail -3 $d.log; exit 1; }
  echo "== $e" | tee -a gpurun_out/${tag}_ab.txt
  python3 tools/trace_summary.py $(ls $d/*kernel_trace.csv | head -1) 40 | grep "$kern\|total ms" | tee -a gpurun_out/${tag}_ab.txt
  rm -f $d/*kernel_trace.csv
done
