#!/bin/bash
# GPU box: the qkv layer-piece tests against every libptts_fc_*.so variant given
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in "$@"; do
  PTTS_LIB_PATH=$GRAFT_REPO_ROOT/tools/probes/mfma_hazard/build/libptts_fc_$v.so timeout -k 10 200 python3 -m pytest tests/test_gpu_mimi_transformer.py -q -p no:cacheprovider -m gpu -k "layer_piece_qkv" > gpurun_out/r5_fc_$v.log 2>&1
  echo "variant $v: rc=$? $(tail -1 gpurun_out/r5_fc_$v.log)"
done
