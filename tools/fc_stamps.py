"""k_flow_cluster under its in-kernel timestamps (PTTS_FC_STAMPS): per phase, how long the sweep waited, the staging, the barrier, the product + reduce and the
publish took -- median over the workgroups of the last launch of a 64-row batch.   python3 tools/fc_stamps.py  (GPU box)"""
import os
import statistics
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = "/tmp/fc_stamps.txt"
if len(sys.argv) < 2:
    if os.path.exists(out):
        os.remove(out)
    env = dict(os.environ, PTTS_FC_STAMPS=out, PTTS_PROBE_STEPS="12", PTTS_PROBE_REPS="2", PTTS_PROBE_GRAPH="0")
    subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "traffic_probe.py")], env=env, check=True)
else:
    out = sys.argv[1]
groups = open(out).read().strip().split("\n\n")
wgs = [[int(x) for x in l.split(":")[1].split()] for l in groups[-1].splitlines() if l.startswith("wg")]
t0 = min(w[0] for w in wgs)
print(f"{len(wgs)} workgroups; launch spread of starts {10*(max(w[0] for w in wgs)-t0)} ns; kernel {10*(max(w[63] for w in wgs)-t0)/1e3:.2f} us")
names = ["sweep+LN done", "staged", "barrier", "mma+reduce+publish", "sweep done", "staged", "barrier", "mma+reduce+publish"]
prev = [w[0] for w in wgs]
for r in range(6):
    for i in range(8):
        idx = 1 + 8 * r + i
        d = [10 * (w[idx] - p) for w, p in zip(wgs, prev)]
        print(f"  block {r} {names[i]:22s} median {statistics.median(d):7.0f} ns  min {min(d):6.0f}  max {max(d):6.0f}   (at {10*(statistics.median(w[idx] for w in wgs)-t0)/1e3:6.2f} us)")
        prev = [w[idx] for w in wgs]
