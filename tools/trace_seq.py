"""Time-ordered kernel list of the tail of a rocprofv3 kernel_trace.csv with durations and the gap to the previous kernel:
    python tools/trace_seq.py <trace.csv> [last_n]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = rows[-n:]
prev_end = None
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ptts::", "").replace("ptts::", "")
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  +{gap:7.1f}  {(e - s) / 1e3:8.1f} us  {name[:60]:60s} grid=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) wg={r['Workgroup_Size_X']}")
    prev_end = max(prev_end or e, e)
