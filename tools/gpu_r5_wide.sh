#!/bin/bash
# GPU box: the wide-batch tests, the neighbouring 64-row tests, then the one-shot probe at 64 / 128 / 256 rows.  usage: tools/gpu_r5_wide.sh TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=${1:-r5wide}
timeout -k 10 900 python3 -m pytest tests/test_gpu_wide_batch.py tests/test_gpu_flow_cluster.py -x -q -p no:cacheprovider -m gpu > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 25 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 tools/big_batch_probe.py > gpurun_out/${tag}_probe.log 2>&1; rc=$?
tail -n 12 gpurun_out/${tag}_probe.log; exit $rc
