"""Summary of a PTTS_CONT_TRACE file (continuous.cpp): where the step stream's time goes, by what ran beside each group of steps."""
import collections
import sys

print("".join(l for l in open(sys.argv[1]) if l.startswith(("# decoded", "# start_decode"))), end="")
rows = [tuple(float(x) for x in l.split()) for l in open(sys.argv[1]) if l.strip() and not l.startswith("#")]
rows = rows[len(rows) // 3:]   # the timed part
cls = collections.defaultdict(lambda: [0, 0.0, 0.0, 0, 0, 0.0, 0.0, 0.0, 0.0])
for row in rows:
    gap, dur, n_gen, steps, adm, dstart, dec, join = row[:8]
    host = row[8:12] if len(row) >= 12 else (0.0, 0.0, 0.0, 0.0)
    key = ("admit " if adm else "") + ("decode-start " if dstart else "") + ("decoding " if dec else "") + ("joining " if join else "") or "alone"
    c = cls[key]
    c[0] += 1; c[1] += gap; c[2] += dur; c[3] += int(steps); c[4] += int(n_gen * steps)
    for k in range(4):
        c[5 + k] += host[k]
tot = sum(c[1] + c[2] for c in cls.values())
nst = sum(c[3] for c in cls.values())
print(f"groups {sum(c[0] for c in cls.values())}, steps {nst}, time {tot/1e3:.1f} ms = {tot/nst:.0f} us per step, mean occupancy {sum(c[4] for c in cls.values())/nst:.1f}")
for k, c in sorted(cls.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"  {k:44s} groups {c[0]:5d}  {100*(c[1]+c[2])/tot:5.1f}% of time  per step: gap {c[1]/c[3]:7.1f} us + run {c[2]/c[3]:7.1f} us  occupancy {c[4]/c[3]:5.1f}"
          f"  | host per group: {c[5]/c[0]:6.0f} us between enqueues (admit {c[6]/c[0]:5.0f}, start_decode {c[7]/c[0]:5.0f}, waiting for a read-back {c[8]/c[0]:5.0f})")
