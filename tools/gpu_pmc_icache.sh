#!/bin/bash
# GPU box: instruction-cache and issue counters of the AR step's kernels (plain launches, tools/traffic_probe.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=${1:-r3pmc}
rocprofv3 -L 2>/dev/null | grep -io "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u > gpurun_out/${tag}_counters.txt
cat gpurun_out/${tag}_counters.txt | tr '\n' ' '; echo
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  n=$(echo $set | cut -d' ' -f1)
  mkdir -p gpurun_out/${tag}_$n
  PTTS_PROBE_STEPS=12 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/${tag}_$n -o pmc -- python3 tools/traffic_probe.py > gpurun_out/${tag}_$n.log 2>&1; echo "pmc $n rc=$?"
  f=$(ls gpurun_out/${tag}_$n/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f k_skinny > gpurun_out/${tag}_$n.txt && python3 tools/pmc_summary.py $f k_attn_step >> gpurun_out/${tag}_$n.txt && rm -f gpurun_out/${tag}_$n/*kernel_trace.csv gpurun_out/${tag}_$n/*counter_collection.csv
  head -40 gpurun_out/${tag}_$n.txt | cut -c1-220
done
