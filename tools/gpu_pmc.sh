#!/bin/bash
# GPU box: one rocprofv3 counter pass over a short batch-64 probe (tools/prefill_probe.py).  Usage: tools/gpu_pmc.sh <tag> <counter> [counter...]
# Counters go in their own run with --kernel-trace only (no sys / hip / hsa tracing: the pool's gpurun refuses that mix).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
PTTS_PROBE_STEPS=${PTTS_PROBE_STEPS:-12} timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/$tag -o pmc -- python3 tools/prefill_probe.py > gpurun_out/$tag.log 2>&1
rc=$?
tail -n 5 gpurun_out/$tag.log | cut -c1-300
echo "rocprofv3 $* rc=$rc"
exit $rc
