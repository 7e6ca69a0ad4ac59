"""k_flow_cluster under its in-kernel timestamps (PTTS_FC_STAMPS): per linear, when the multiplying waves passed the "image complete" barrier and had
published, and when a staging wave held its row and had staged it -- medians over the workgroups of the last launch of a 64-row batch.
    python3 tools/fc_stamps.py  (GPU box)"""
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
wgs = [w for w in wgs if w[32]]   # (workgroups whose staging wave 4 has a row)
t0 = min(w[0] for w in wgs)
med = lambda i: 10 * (statistics.median(w[i] for w in wgs) - t0) / 1e3
print(f"{len(wgs)} workgroups; spread of starts {10*(max(w[0] for w in wgs)-t0)} ns; kernel {10*(max(w[31] for w in wgs)-t0)/1e3:.2f} us")
print("times in us from the first workgroup's start (medians over the workgroups)")
print("block | x row held | staged | barrier passed | h published || h row held | staged | barrier passed | x published")
for r in range(6):
    print(f"  {r}   | {med(32+4*r):8.2f}   | {med(33+4*r):6.2f} | {med(1+4*r):8.2f}       | {med(2+4*r):8.2f}    || {med(34+4*r):8.2f}   | {med(35+4*r):6.2f} | {med(3+4*r):8.2f}       | {med(4+4*r):8.2f}")
hop = [med(34 + 4 * r) - med(2 + 4 * r) for r in range(6)] + [med(32 + 4 * r) - med(4 + 4 * (r - 1)) for r in range(1, 6)]
work = [med(2 + 4 * r) - med(33 + 4 * r) for r in range(6)] + [med(4 + 4 * r) - med(35 + 4 * r) for r in range(6)]
print(f"published -> row held by the stager (the hop): median {statistics.median(hop):.2f} us; staged -> published (barrier, products, epilogue): median {statistics.median(work):.2f} us")
b = lambda i: 10 * statistics.median(w[i] - w[1 + 4 * 2] for w in wgs)
print(f"block 2, mlp0, multiplying wave 0 after the barrier: sums final +{b(56):.0f} ns, next weights requested +{b(57):.0f}, SiLU done +{b(58):.0f}, published +{b(2 + 4 * 2):.0f}")
