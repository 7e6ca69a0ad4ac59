#!/bin/bash
# GPU box: the shipped library, then the ablation build.  usage: tools/probes/hi_only/run.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{ echo "== shipped"; timeout -k 10 500 python3 tools/probes/hi_only/measure.py; echo "== hi only (-DPTTS_ABLATE_LO)"; PTTS_LIB_PATH=$GRAFT_REPO_ROOT/tools/probes/hi_only/build/libptts_hip.so timeout -k 10 500 python3 tools/probes/hi_only/measure.py; } 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/hi_only.txt
