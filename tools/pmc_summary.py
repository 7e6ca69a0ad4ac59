"""Per-kernel means of a rocprofv3 counter_collection.csv: python tools/pmc_summary.py file.csv [kernel substring]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for r in rows:
    if sub not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"][:48], r["Grid_Size"])
    d = agg.setdefault(k, collections.OrderedDict())
    d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for (kn, g), d in agg.items():
    print(kn, "grid", g, "n", len(next(iter(d.values()))))
    print("   " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in d.items()))
