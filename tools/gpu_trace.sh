#!/bin/bash
# GPU box: rocprofv3 kernel trace of the short batch-64 probe, per-(kernel, grid) table.  Usage: tools/gpu_trace.sh <tag> [env assignments...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
env "$@" PTTS_PROBE_STEPS=${PTTS_PROBE_STEPS:-25} true
for kv in "$@"; do export "$kv"; done
export PTTS_PROBE_STEPS=${PTTS_PROBE_STEPS:-25}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o tr -- python3 tools/traffic_probe.py > gpurun_out/$tag.log 2>&1
echo "rocprofv3 rc=$?"
python3 tools/trace_summary.py $(ls gpurun_out/$tag/*kernel_trace.csv | head -1) 45 > gpurun_out/${tag}_by_grid.txt
cat gpurun_out/${tag}_by_grid.txt
