"""Timeline of a serving run from a rocprofv3 kernel trace (tools/gpu_serve_trace.sh): how long an AR step of the continuous engine takes
alone, beside a decode (second stream), beside a prefill (I/O stream), and how long the step queue sat empty.
    python3 tools/serve_timeline.py <kernel_trace.csv>"""
import bisect
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ptts::", "").replace("ptts::", "")
    q = r.get("Queue_Id") or r.get("Stream_Id") or "0"
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), q, name))
rows.sort()


def is_step_end(name):   # k_skinny<WT, STAMP, PRO, NJ, CG, FIN = true, ...>  or the stand-alone bookkeeping kernel
    if name.startswith("k_step_finish"):
        return True
    if not name.startswith("k_skinny<"):
        return False
    args = [a.strip() for a in name[len("k_skinny<"):name.index(">")].split(",")]
    return len(args) >= 6 and args[5] == "true"


per_q = collections.defaultdict(list)
for s, e, q, n in rows:
    per_q[q].append((s, e, n))
ends_by_q = {q: sum(is_step_end(n) for _, _, n in v) for q, v in per_q.items()}
step_q = max(ends_by_q, key=ends_by_q.get)
print("queues:", {q: (len(v), ends_by_q[q]) for q, v in per_q.items()}, "step queue:", step_q)


def classify(n):
    if n.startswith(("k_mimi", "k_resblock", "k_gemm5", "k_gemm_wres", "k_attn_window", "k_upsample", "k_projector", "k_conv", "k_pcm", "k_gather_frames")):
        return "decode"
    return "other"


# busy intervals of everything that is not on the step queue, by what it is
other = {"decode": [], "other": []}
for q, v in per_q.items():
    if q == step_q:
        continue
    for s, e, n in v:
        other[classify(n)].append((s, e))
for k in other:
    other[k].sort()
    merged = []
    for s, e in other[k]:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    other[k] = merged


def overlap(iv, a, b):
    i = bisect.bisect_left(iv, [a, a]) - 1
    tot = 0
    for s, e in iv[max(i, 0):]:
        if s >= b:
            break
        tot += max(0, min(e, b) - max(s, a))
    return tot


sq = per_q[step_q]
ends = [i for i, (_, _, n) in enumerate(sq) if is_step_end(n)]
# the timed part: the last 55 % of the steps (warm-up pass and the untimed round come first)
ends = ends[int(0.45 * len(ends)):]
cls = collections.defaultdict(lambda: [0, 0, 0, 0])   # class -> [steps, period ns, kernel ns inside, idle ns inside]
idle_hist = collections.Counter()
for a, b in zip(ends[:-1], ends[1:]):
    t0, t1 = sq[a][1], sq[b][1]
    per = t1 - t0
    if per > 20_000_000:
        continue
    busy = sum(e - s for s, e, _ in sq[a + 1:b + 1])
    big_idle = 0
    prev = t0
    for s, e, _ in sq[a + 1:b + 1]:
        if s - prev > 20_000:
            big_idle += s - prev
            idle_hist[min(10, (s - prev) // 100_000)] += 1
        prev = max(prev, e)
    od, oo = overlap(other["decode"], t0, t1) / per, overlap(other["other"], t0, t1) / per
    has_pref = any(not is_step_end(n) and not n.startswith(("k_skinny", "k_attn_step", "k_step")) for _, _, n in sq[a + 1:b + 1])
    key = ("decode " if od > 0.3 else "") + ("io " if oo > 0.3 else "") + ("inline-extra " if has_pref else "")
    c = cls[key or "alone"]
    c[0] += 1; c[1] += per; c[2] += busy; c[3] += big_idle
tot = sum(c[1] for c in cls.values())
print(f"steps analysed {sum(c[0] for c in cls.values())}, time {tot/1e6:.1f} ms")
for k, c in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} steps {c[0]:6d}  {100*c[1]/tot:5.1f}% of time  period {c[1]/c[0]/1e3:8.1f} us  kernels {c[2]/c[0]/1e3:8.1f} us  gaps>20us {c[3]/c[0]/1e3:8.1f} us")
print("idle gaps on the step queue by length (x100 us):", dict(sorted(idle_hist.items())))
# what else ran on the step queue between steps (admission writes, ...)
extra = collections.defaultdict(lambda: [0, 0])
for a, b in zip(ends[:-1], ends[1:]):
    for s, e, n in sq[a + 1:b + 1]:
        if not n.startswith(("k_skinny", "k_attn_step", "k_step")):
            extra[n.split("(")[0][:50]][0] += 1; extra[n.split("(")[0][:50]][1] += e - s
for k, v in sorted(extra.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  on the step queue: {k:50s} n={v[0]:6d} total {v[1]/1e6:8.2f} ms")
