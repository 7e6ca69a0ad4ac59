#!/bin/bash
# GPU box: continuous serving at PTTS_SLOTS / CLIENTS of each setting with the engine's group trace summarised.  usage: tools/gpu_cont_trace2.sh TAG "ENV... CLIENTS=n" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1; shift
: > gpurun_out/${tag}.txt
i=0
for e in "$@"; do
  i=$((i+1)); rm -f /tmp/ct_$i.txt
  clients=$(echo "$e" | tr ' ' '\n' | grep '^CLIENTS=' | cut -d= -f2)
  envs=$(echo "$e" | tr ' ' '\n' | grep -v '^CLIENTS=' | tr '\n' ' ')
  echo "== $envs" | tee -a gpurun_out/${tag}.txt
  env $envs PTTS_CONTINUOUS=1 PTTS_CONT_TRACE=/tmp/ct_$i.txt PTTS_ENGINES=1 PTTS_PER_CLIENT=4 timeout -k 10 240 python3 tools/serve_bench.py $clients 2>&1 | grep "x real time" | tee -a gpurun_out/${tag}.txt
  python3 tools/cont_trace_summary.py /tmp/ct_$i.txt | tee -a gpurun_out/${tag}.txt
done
