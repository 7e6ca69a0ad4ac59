#!/bin/bash
# GPU box: the headline (quick form) under several environment settings, in rotation.  usage: tools/gpu_ab_multi.sh TAG ROUNDS "ENV1" "ENV2" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1; n=$2; shift 2
for r in $(seq 1 $n); do
  for e in "$@"; do
    env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-b1 --no-traffic --no-two-engines --steps 8 > gpurun_out/${tag}_ab.json 2> gpurun_out/${tag}_ab.err || { echo "$e failed"; tail -3 gpurun_out/${tag}_ab.err; exit 1; }
    python3 -c "import json;d=json.load(open('gpurun_out/${tag}_ab.json'));r=d['roofline'];print('$e', d['value'],d['ms_per_step'],r['avg_launch_us'],r.get('phases_ms'))" | tee -a gpurun_out/${tag}_ab.txt
  done
done
