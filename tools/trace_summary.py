"""Summarise a rocprofv3 kernel_trace.csv by (kernel, grid): python tools/trace_summary.py <trace.csv> [top]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ptts::", "").replace("ptts::", "")
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key][0] += 1
    agg[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0]:30s} grid=({k[1]},{k[2]},{k[3]}) n={v[0]:6d} total={v[1]/1e6:9.2f} ms avg={v[1]/v[0]/1e3:9.1f} us {100*v[1]/tot:5.1f}%")
print("total ms", tot / 1e6)
