#!/bin/bash
# GPU box: tools/serve_bench.py under a list of environment settings.  usage: tools/gpu_serve_sweep.sh TAG "ENV... CLIENTS=a,b" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1; shift
for e in "$@"; do
  clients=$(echo "$e" | tr ' ' '\n' | grep '^CLIENTS=' | cut -d= -f2 | tr ',' ' ')
  envs=$(echo "$e" | tr ' ' '\n' | grep -v '^CLIENTS=' | tr '\n' ' ')
  echo "== $envs" | tee -a gpurun_out/${tag}_serve.txt
  env $envs timeout -k 10 400 python3 tools/serve_bench.py $clients 2> gpurun_out/${tag}_serve.err | tee -a gpurun_out/${tag}_serve.txt || { echo "failed: $e"; tail -5 gpurun_out/${tag}_serve.err; exit 1; }
done
