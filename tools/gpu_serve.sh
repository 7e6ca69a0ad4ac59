#!/bin/bash
# GPU box: the serving benchmark, batch-at-a-time against continuous batching, uniform 10-s and mixed 2-12-s utterances, one and two engines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=${1:-r3serve}; shift
clients=${*:-"64 128"}
: > gpurun_out/${tag}.txt
for eng in 1 2; do for mixed in 0 1; do for cont in 0 1; do
  PTTS_ENGINES=$eng PTTS_MIXED=$mixed PTTS_CONTINUOUS=$cont PTTS_PER_CLIENT=4 timeout -k 10 240 python3 tools/serve_bench.py $clients 2>&1 | grep "x real time" | tee -a gpurun_out/${tag}.txt
done; done; done
