#!/bin/bash
# GPU box: k_gemm5 against k_gemm3 bit for bit, every epilogue form the decoder uses, and against itself over repeated runs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1
timeout -k 10 600 python3 tools/microbench_gemm.py ${2:-0.5} 50 qkv,qkv_rope,out_proj_ip,ffn2_ip,ffn1,init_conv,up1,rb1_0_elu,rb2_0_re,rb1_0,rb2_0 > gpurun_out/${tag}_g5c.txt 2>&1; echo "check rc=$?"
timeout -k 10 600 python3 tools/microbench_gemm.py ${2:-0.5} 52 qkv,qkv_rope,out_proj_ip,ffn2_ip,up1 >> gpurun_out/${tag}_g5c.txt 2>&1; echo "check rc=$?"
cat gpurun_out/${tag}_g5c.txt
