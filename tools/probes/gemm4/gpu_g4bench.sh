#!/bin/bash
# GPU box: the headline with k_gemm4 off / persistent / two-stage (PTTS_GEMM4 = 0 / 1 / 2), phases
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=${1:-r2g4}
for v in 0 1 2 0 1 2; do
  PTTS_GEMM4=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-b1 --no-traffic --no-two-engines --steps 8 > gpurun_out/${tag}_bench_$v.json 2> gpurun_out/${tag}_bench_$v.err || { echo "bench $v failed"; tail -5 gpurun_out/${tag}_bench_$v.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/${tag}_bench_$v.json'));print('PTTS_GEMM4=$v', d['value'],d['ms_per_step'],d['roofline']['phases_ms'])" | tee -a gpurun_out/${tag}_summary.txt
done
