"""does stepping the 64 utterances of ONE batch as two (or four) independent sub-batches on their own streams
beat stepping them together?  Emulated exactly with engines that share the weight arena (ptts_model_share: own streams, KV caches,
workspaces): k engines x 64/k utterances each, run concurrently from k threads, against 1 engine x 64.  Same total work, same
utterances in flight; what changes is how many dependent launch chains overlap.
    python tools/halves_probe.py"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (shares the HIP runtime)
import bench
import ptts_amd

pkg = ptts_amd.load()
wl = bench.WORKLOADS["b64_10s_bf16"]
cfg = pkg.synth.SynthConfig.full()
path = bench.checkpoint_path(pkg, wl["file"], 0, lambda: None)
graph = os.environ.get("PTTS_PROBE_GRAPH", "1") != "0"
base = pkg.Model.open(path, device=0, weights=wl["weights"], kv=wl["kv"], max_batch=64, use_graph=graph)
voice = base.upload_voice(pkg.VoiceModelState(bench.voice_modules(pkg, cfg)))
prompts = [np.ascontiguousarray(p, np.int64) for p in pkg.synth.make_prompts(64, 25, 4000, seed=42)]
engines = [base] + [base.share() for _ in range(3)]
for e in engines[1:]:
    e.set_use_graph(graph)


def run(k, reps=6):
    per = 64 // k
    parts = [prompts[i * per:(i + 1) * per] for i in range(k)]
    cfgs = bench.gen_cfgs(pkg, wl, per, voice)

    def work(e, toks):
        e.generate_batch(toks, cfgs)

    for _ in range(2):   # warm-up (workspaces, graphs)
        ts = [threading.Thread(target=work, args=(engines[i], parts[i])) for i in range(k)]
        [t.start() for t in ts]; [t.join() for t in ts]
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(engines[i], parts[i])) for i in range(k)]
        [t.start() for t in ts]; [t.join() for t in ts]
        times.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(times))


for k in (1, 2, 4, 1, 2, 4):
    ms = run(k)
    print(f"{k} engine(s) x {64 // k:2d} utterances, graph={int(graph)}: {ms:7.2f} ms per 64 utterances  ({64 * 10.0 / (ms * 1e-3):8.0f} x real time)", flush=True)
