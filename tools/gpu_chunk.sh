#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for c in 0 100 64 50 25 0; do
  if [ $c -eq 0 ]; then unset PTTS_MIMI_CHUNK; else export PTTS_MIMI_CHUNK=$c; fi
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-b1 --no-traffic --no-two-engines --steps 8 > gpurun_out/r2n_chunk_$c.json 2> gpurun_out/r2n_chunk_$c.err
  python3 -c "import json;d=json.load(open('gpurun_out/r2n_chunk_$c.json'));print('PTTS_MIMI_CHUNK=$c', d['value'], d['ms_per_step'])"
done
