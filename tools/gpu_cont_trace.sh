#!/bin/bash
# GPU box: continuous serving (mixed 2-12 s, 128 clients, one engine) under settings, each with the engine's group trace summarised.
# usage: tools/gpu_cont_trace.sh TAG "ENV..." "ENV..." ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1; shift
: > gpurun_out/${tag}.txt
i=0
for e in "$@"; do
  i=$((i+1)); rm -f /tmp/ct_$i.txt
  echo "== $e" | tee -a gpurun_out/${tag}.txt
  env $e PTTS_CONTINUOUS=1 PTTS_CONT_TRACE=/tmp/ct_$i.txt PTTS_ENGINES=1 PTTS_MIXED=1 PTTS_PER_CLIENT=4 timeout -k 10 240 python3 tools/serve_bench.py 128 2>&1 | grep "x real time" | tee -a gpurun_out/${tag}.txt
  python3 tools/cont_trace_summary.py /tmp/ct_$i.txt | tee -a gpurun_out/${tag}.txt
done
