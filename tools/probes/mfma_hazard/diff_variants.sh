#!/bin/bash
# GPU box: where does a variant's k_mimi_rowlin differ from the shipped library's?  usage: diff_variants.sh ROWS VARIANT...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
rows=$1; shift
OUT=/tmp/ref.npy timeout -k 5 120 python3 tools/probes/rowlin_debug.py $rows > /dev/null 2>&1 || { echo "reference run failed"; exit 1; }
for v in "$@"; do
  echo "== variant $v rows=$rows"
  REF=/tmp/ref.npy OUT=/tmp/v.npy PTTS_LIB_PATH=$GRAFT_REPO_ROOT/tools/probes/mfma_hazard/build/libptts_fc_$v.so timeout -k 5 120 python3 tools/probes/rowlin_debug.py $rows 2>&1 | tail -5
done
