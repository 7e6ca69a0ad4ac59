#!/bin/bash
# GPU box: continuous batching under tuning settings (mixed 2-12 s utterances, one engine).  usage: tools/gpu_serve_sweep.sh TAG CLIENTS "ENV..." "ENV..." ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1; clients=$2; shift 2
: > gpurun_out/${tag}.txt
for e in "$@"; do
  echo "== $e" | tee -a gpurun_out/${tag}.txt
  env $e PTTS_ENGINES=${PTTS_ENGINES:-1} PTTS_MIXED=${PTTS_MIXED:-1} PTTS_PER_CLIENT=4 timeout -k 10 240 python3 tools/serve_bench.py $clients 2>&1 | grep "x real time" | tee -a gpurun_out/${tag}.txt
done
