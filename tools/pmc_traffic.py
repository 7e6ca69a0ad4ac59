"""profiles/pmc_traffic.json from a rocprofv3 --pmc FETCH_SIZE pass over bench.py.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/*/*counter_collection.csv b64_10s_bf16

FETCH_SIZE is in KiB and, on gfx950, reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section:
128-byte requests tallied at 64 B): the value is doubled.  The counter sits on the L2's fabric side, so per-XCD refills of
data another XCD produced (activations, LayerNorm vectors) are in it -- eight L2s each fetch their own copy.
"""
import collections
import csv
import json
import os
import sys

rows = csv.DictReader(open(sys.argv[1]))
wl = sys.argv[2] if len(sys.argv) > 2 else "b64_10s_bf16"
per = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    name = r["Kernel_Name"]
    base = name.split("(")[0].replace("void ", "").split("<")[0].split("::")[-1].strip()
    per[base][0] += float(r["Counter_Value"]) * 1024.0 * 2.0
    per[base][1] += 1
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "pmc_traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}
for k, (b, n) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    out.setdefault(k, {})[wl] = round(b / n)
    print(f"{k:24s} launches {n:6d}  fetched per launch {b/n/1e6:9.3f} MB  total {b/1e9:7.2f} GB")
out["_note"] = ("bytes per launch = FETCH_SIZE (KiB) x 1024 x 2 (gfx950 correction), mean over every dispatch of the kernel in one bench.py pass; "
                "reads only: the WRITE_SIZE pass crashes rocprofv3 on this image")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
