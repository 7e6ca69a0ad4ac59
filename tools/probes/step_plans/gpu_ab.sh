#!/bin/bash
# GPU box: tests, then the headline with the old and the new step plan (A/B), then in-situ stamps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=${1:-r2g}
tools/gpu_tests.sh $tag; rc=$?
if [ $rc -gt 1 ]; then exit $rc; fi
for v1 in 1 0 2 1 0 2; do
  PTTS_STEP_PLAN=$v1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-b1 --no-traffic --no-two-engines --steps 8 > gpurun_out/${tag}_bench_v1_$v1.json 2> gpurun_out/${tag}_bench_v1_$v1.err
  echo "bench PTTS_STEP_PLAN=$v1 rc=$?"; python3 -c "import json;d=json.load(open('gpurun_out/${tag}_bench_v1_$v1.json'));print(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['launches'],d['roofline']['phases_ms'])"
done
timeout -k 10 200 python3 tools/step_stamps.py > gpurun_out/${tag}_step_stamps.txt 2>&1; echo "stamps rc=$?"
exit $rc
