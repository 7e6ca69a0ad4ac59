#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1; rm -f gpurun_out/${tag}_g5s.txt
for st in 0 2000 3200 5000; do
  echo "## PTTS_GEMM5_STAGGER=$st" >> gpurun_out/${tag}_g5s.txt
  PTTS_GEMM5_STAGGER=$st timeout -k 10 300 python3 tools/microbench_gemm.py 1.0 51,52 qkv,qkv_rope,out_proj,ffn2,up1 >> gpurun_out/${tag}_g5s.txt 2>&1 || exit 1
done
grep -v "amdgpu\|vs reference" gpurun_out/${tag}_g5s.txt
