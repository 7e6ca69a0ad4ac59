#!/bin/bash
# GPU box: k_gemm5 against k_gemm3 / the dispatch default on the decoder's shapes, then the ablation build.  usage: tools/gpu_g5.sh TAG [variants] [shapes]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1; vars=${2:-3,40,51,52,53,54}; shapes=${3:-qkv,out_proj,ffn1,ffn2,init_conv,up1,rb1_0,rb2_0}
timeout -k 10 400 python3 tools/microbench_gemm.py 1.0 $vars $shapes > gpurun_out/${tag}_g5.txt 2>&1; echo "microbench rc=$?"
cat gpurun_out/${tag}_g5.txt
if [ -f go-pocket-tts_amd/ab/libptts_probe.so ] && [ -n "$4" ]; then
  PTTS_LIB_PATH="$GRAFT_REPO_ROOT/go-pocket-tts_amd/ab/libptts_probe.so" timeout -k 10 300 python3 tools/microbench_gemm.py 1.0 $4 ${5:-qkv,ffn2} > gpurun_out/${tag}_g5_abl.txt 2>&1; echo "ablation rc=$?"
  cat gpurun_out/${tag}_g5_abl.txt
fi
