#!/bin/bash
# GPU box: AR-step diagnostics of round 2 (stamps, per-shape launch times, launch floor, kernarg placement A/B, rocprof stats)
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
hipcc -O3 --offload-arch=gfx950 -o /tmp/launch_floor tools/probes/launch_floor.hip && timeout -k 10 120 /tmp/launch_floor > gpurun_out/r2c_launch_floor.txt 2>&1
echo "launch_floor rc=$?"; cat gpurun_out/r2c_launch_floor.txt
timeout -k 10 200 python3 tools/stamps_skinny.py > gpurun_out/r2c_stamps.txt 2>&1; echo "stamps rc=$?"
timeout -k 10 300 python3 tools/microbench.py > gpurun_out/r2c_microbench.txt 2>&1; echo "microbench rc=$?"
for kv in 0 1; do
  HIP_FORCE_DEV_KERNARG=$kv timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-b1 --no-two-engines --no-traffic --steps 5 > gpurun_out/r2c_bench_devkernarg$kv.json 2> gpurun_out/r2c_bench_devkernarg$kv.err
  echo "bench HIP_FORCE_DEV_KERNARG=$kv rc=$?"; python3 -c "import json;d=json.load(open('gpurun_out/r2c_bench_devkernarg$kv.json'));print(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['phases_ms'])"
done
