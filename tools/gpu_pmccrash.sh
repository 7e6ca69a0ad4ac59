#!/bin/bash
# Does the round-1 rocprofv3 crash (--pmc WRITE_SIZE over bench.py) come from hipGraph capture under counter collection?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r2r_a gpurun_out/r2r_b
PTTS_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2r_a -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-b1 --no-two-engines --no-traffic > gpurun_out/r2r_plain.log 2>&1
echo "plain launches under --pmc WRITE_SIZE: rc=$?"; tail -n 3 gpurun_out/r2r_plain.log | cut -c1-200
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2r_b -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-b1 --no-two-engines --no-traffic > gpurun_out/r2r_graph.log 2>&1
echo "graph capture + replay under --pmc WRITE_SIZE: rc=$?"; grep -n "SIGSEGV\|Aborted\|launch_attn_step\|step_graph\|hipStreamBeginCapture\|hipGraph" gpurun_out/r2r_graph.log | head -8; tail -n 3 gpurun_out/r2r_graph.log | cut -c1-200
