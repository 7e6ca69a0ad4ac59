#!/bin/bash
# GPU box: selected tests under an environment.  usage: tools/gpu_one.sh TAG "ENV=..." pytest-args...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1; envs=$2; shift 2
env $envs timeout -k 10 900 python3 -m pytest "$@" -x -q -p no:cacheprovider > gpurun_out/${tag}_one.log 2>&1; rc=$?
tail -n 15 gpurun_out/${tag}_one.log; exit $rc
