#!/bin/bash
# GPU box: where the GPU's time goes while the dispatcher serves (kernel trace of tools/serve_bench.py, totals per kernel name).
# usage: tools/gpu_serve_trace.sh TAG CLIENTS "ENV..."
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/${1}_tr
env $3 PTTS_PER_CLIENT=3 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${1}_tr -o tr -- python3 tools/serve_bench.py $2 > gpurun_out/${1}_tr.log 2>&1 || { tail -3 gpurun_out/${1}_tr.log; exit 1; }
grep "x real time" gpurun_out/${1}_tr.log
python3 - "$(ls gpurun_out/${1}_tr/*kernel_trace.csv | head -1)" > gpurun_out/${1}_serve_by_kernel.txt <<'PY'
import collections, csv, sys
agg = collections.defaultdict(lambda: [0, 0])
t0, t1 = None, None
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ptts::", "").replace("ptts::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    agg[name][0] += 1; agg[name][1] += e - s
    t0 = s if t0 is None else min(t0, s); t1 = e if t1 is None else max(t1, e)
tot = sum(v[1] for v in agg.values())
print(f"wall {1e-6*(t1-t0):.1f} ms, sum of kernel durations {1e-6*tot:.1f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k[:60]:60s} n={v[0]:8d} total={v[1]/1e6:10.2f} ms avg={v[1]/v[0]/1e3:9.1f} us {100*v[1]/tot:5.1f}%")
PY
cat gpurun_out/${1}_serve_by_kernel.txt
python3 tools/serve_timeline.py "$(ls gpurun_out/${1}_tr/*kernel_trace.csv | head -1)" > gpurun_out/${1}_serve_timeline.txt 2>&1; cat gpurun_out/${1}_serve_timeline.txt
rm -f gpurun_out/${1}_tr/*kernel_trace.csv
