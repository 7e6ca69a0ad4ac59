#!/bin/bash
# GPU box: the -m gpu suite in one process, log + observed parity errors under gpurun_out/.  Usage: tools/gpu_tests.sh [tag] [pytest args...]
tag=${1:-r2}; shift
mkdir -p gpurun_out
rm -f gpurun_out/parity_observed.jsonl
timeout -k 10 1000 python3 -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider "$@" > gpurun_out/${tag}_pytest.log 2>&1
rc=$?
tail -n 25 gpurun_out/${tag}_pytest.log
cp -f gpurun_out/parity_observed.jsonl gpurun_out/${tag}_parity_observed.jsonl 2>/dev/null
echo "pytest rc=$rc"
exit $rc
