#!/bin/bash
# GPU box: k_gemm4 against k_gemm3 / the dispatcher's choice on the Mimi decoder's deep shapes (tools/microbench_gemm.py)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=${1:-r2g4}
shapes=qkv,out_proj,ffn1,ffn2,init_conv,up1
timeout -k 10 300 python3 tools/microbench_gemm.py 0.25 3,50,51 $shapes > gpurun_out/${tag}_quarter.txt 2>&1 || { tail -5 gpurun_out/${tag}_quarter.txt; exit 1; }
cat gpurun_out/${tag}_quarter.txt
timeout -k 10 400 python3 tools/microbench_gemm.py 1.0 3,50,51 $shapes > gpurun_out/${tag}_full.txt 2>&1 || { tail -5 gpurun_out/${tag}_full.txt; exit 1; }
cat gpurun_out/${tag}_full.txt
