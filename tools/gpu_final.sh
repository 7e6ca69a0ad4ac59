#!/bin/bash
# GPU box: the round's evidence on the final build -- tests, bench (the driver's command), rocprofv3 kernel stats of the same command,
# SQ counters of the AR step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=${1:-r2final}
tools/gpu_tests.sh $tag; rc=$?
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/${tag}_bench.json; echo
mkdir -p gpurun_out/${tag}_prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o prof -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-two-engines --no-traffic > gpurun_out/${tag}_prof.log 2>&1; echo "rocprof rc=$?"
python3 tools/trace_summary.py $(ls gpurun_out/${tag}_prof/*kernel_trace.csv | head -1) 70 > gpurun_out/${tag}_by_grid.txt; head -30 gpurun_out/${tag}_by_grid.txt
mkdir -p gpurun_out/${tag}_sq
PTTS_PROBE_STEPS=12 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${tag}_sq -o sq -- python3 tools/traffic_probe.py > gpurun_out/${tag}_sq.log 2>&1; echo "sq rc=$?"
python3 tools/pmc_summary.py $(ls gpurun_out/${tag}_sq/*counter_collection.csv | head -1) k_skinny > gpurun_out/${tag}_pmc_ar_sq.txt
python3 tools/pmc_summary.py $(ls gpurun_out/${tag}_sq/*counter_collection.csv | head -1) k_attn_step >> gpurun_out/${tag}_pmc_ar_sq.txt
head -12 gpurun_out/${tag}_pmc_ar_sq.txt
exit $rc
