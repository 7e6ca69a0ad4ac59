#!/bin/bash
# GPU box: HBM traffic of the AR step's kernels PER VARIANT (FETCH_SIZE and WRITE_SIZE in passes of their own, plain launches, tools/traffic_probe.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=${1:-r3traffic}
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p gpurun_out/${tag}_$c
  PTTS_PROBE_STEPS=12 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_$c -o pmc -- python3 tools/traffic_probe.py > gpurun_out/${tag}_$c.log 2>&1; echo "pmc $c rc=$?"
  f=$(ls gpurun_out/${tag}_$c/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f k_skinny > gpurun_out/${tag}_$c.txt && python3 tools/pmc_summary.py $f k_attn_step >> gpurun_out/${tag}_$c.txt && rm -f gpurun_out/${tag}_$c/*kernel_trace.csv gpurun_out/${tag}_$c/*counter_collection.csv
  cat gpurun_out/${tag}_$c.txt | cut -c1-160
done
