#!/bin/bash
# GPU box: the probe plain and under rocprofv3 --kernel-trace (Queue_Id per kernel k<stream>), with 1 and 2 plain streams beside the masked one
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r5_stream_queues.txt; : > $out
for cfg in "1 0" "2 0" "3 0" "2 1"; do
  for hwq in "" "GPU_MAX_HW_QUEUES=8"; do
    echo "== extra plain streams / masked first: $cfg  $hwq" >> $out
    env $hwq timeout -k 5 120 tools/probes/stream_queues/build/probe $cfg >> $out 2>&1
    d=gpurun_out/sq_trace; rm -rf $d; mkdir -p $d
    env $hwq timeout -k 5 120 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- tools/probes/stream_queues/build/probe $cfg > /dev/null 2>&1
    python3 - $d >> $out <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
q = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    name = r["Kernel_Name"]
    if name.startswith("void k<") or name.startswith("k<"):
        q[name.split("(")[0]].add(r["Queue_Id"])
for k in sorted(q): print("   trace:", k, "-> queue id(s)", sorted(q[k]))
PY
    rm -rf $d
  done
done
cat $out
