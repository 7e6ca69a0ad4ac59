"""Serving benchmark of the dispatcher: closed-loop clients, each synthesising 10-s utterances back to
back through Dispatcher.generate; reports aggregate xRT, utterance latency and the batch sizes the dispatcher formed.

    python tools/serve_bench.py [clients ...]        e.g.  python tools/serve_bench.py 1 8 64 128
Environment: PTTS_ENGINES (engines per GPU), PTTS_WINDOW_US, PTTS_PER_CLIENT, PTTS_CONTINUOUS=1 (continuous batching),
PTTS_MIXED=1 (utterances of 2-12 s, uniformly drawn per request, instead of 10 s each: what EOS does to real traffic),
PTTS_SLOTS (utterances per engine, default 64).
"""
import os
import statistics
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process, see runtime.py lib())
import bench
import ptts_amd

pkg = ptts_amd.load()
wl = bench.WORKLOADS["b64_10s_bf16"]
cfg = pkg.synth.SynthConfig.full()
path = bench.checkpoint_path(pkg, wl["file"], 0, lambda: None)
# PTTS_ENGINES=2: two engines (KV caches, workspaces, streams) over ONE weight arena on the same GPU: batch k+1's prefill and AR
# loop (latency-bound, most of the chip idle) run beside batch k's Mimi decode (throughput work)
n_eng = int(os.environ.get("PTTS_ENGINES", "1"))
SLOTS = int(os.environ.get("PTTS_SLOTS", "64"))   # utterances per engine: the batch collector's max_batch / the continuous engine's slots (up to 256)
kw = dict(device=0, weights=wl["weights"], kv=wl["kv"], max_batch=SLOTS)
models = [pkg.Model.open(path, **kw)]
models += [models[0].share() for _ in range(n_eng - 1)]
model = models[0]
voices = [m.upload_voice(pkg.VoiceModelState(bench.voice_modules(pkg, cfg))) for m in models]
voice = voices[0]
prompts = [p.tolist() for p in pkg.synth.make_prompts(256, 25, 4000, seed=3)]
FRAMES, PER_CLIENT = 125, int(os.environ.get("PTTS_PER_CLIENT", "3"))
window_us = int(os.environ.get("PTTS_WINDOW_US", "3000"))
continuous = os.environ.get("PTTS_CONTINUOUS", "0") == "1"
mixed = os.environ.get("PTTS_MIXED", "0") == "1"
import random
lens = random.Random(5)
frame_plan = [lens.randint(25, 150) if mixed else FRAMES for _ in range(4096)]
# every engine once at full batch before anything is timed: its first call allocates ~16 GB of decoder workspace, the KV caches
# and the page-locked result pool (~0.5 s)
for m_, v_ in zip(models, voices):
    wc = pkg.RuntimeGenerateConfig(max_steps=FRAMES, eos_threshold=float("inf"), frames_after_eos=3, device_voice=v_, pcm16=True)
    m_.generate_batch(prompts[:SLOTS], [wc] * SLOTS)
for clients in [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 128]:
    disp = pkg.Dispatcher(models, max_batch=SLOTS, window_us=window_us, continuous=continuous, cont_kv_capacity=512, cont_max_steps=256,
                          cont_steps_per_group=int(os.environ.get("PTTS_CONT_GROUP", "0")))
    lat, frames_done = [], []
    lock = threading.Lock()

    def client(i):
        for k in range(PER_CLIENT):
            nf = frame_plan[(i * PER_CLIENT + k) % len(frame_plan)]
            c = pkg.RuntimeGenerateConfig(max_steps=nf, eos_threshold=float("inf"), frames_after_eos=3, device_voice=voice, pcm16=True)   # one voice: every engine of this GPU can read it
            t0 = time.perf_counter()
            r = disp.generate(prompts[(i * PER_CLIENT + k) % len(prompts)], c)
            dt = time.perf_counter() - t0
            assert r.n_frames == nf
            with lock:
                lat.append(dt)
                frames_done.append(nf)

    # one untimed round through the dispatcher (first-use allocations of this path: result buffers, decoder workspaces), then the timed rounds
    PER_CLIENT, keep = 1, PER_CLIENT
    ts = [threading.Thread(target=client, args=(i,)) for i in range(clients)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    PER_CLIENT = keep
    lat.clear(); frames_done.clear()
    st0 = disp.stats()
    ts = [threading.Thread(target=client, args=(i,)) for i in range(clients)]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    wall = time.perf_counter() - t0
    st = disp.stats()
    audio = sum(frames_done) * bench.FRAME_SEC
    lat.sort()
    print(f"engines {n_eng} {'continuous' if continuous else 'batch-at-a-time'} {'mixed 2-12 s' if mixed else '10 s'} clients {clients:4d}  window {window_us} us  {audio/wall:8.1f} x real time  latency p50 {1e3*statistics.median(lat):7.1f} ms  "
          f"p95 {1e3*lat[int(0.95*(len(lat)-1))]:7.1f} ms  batches {st['batches'] - st0['batches']:3d}  mean batch {(st['requests'] - st0['requests']) / max(1, st['batches'] - st0['batches']):5.1f}  "
          f"mean queue wait {st['mean_wait_us']/1e3:6.1f} ms"
          + (f"  steps {st['cont_steps'] - st0['cont_steps']} at {1e6*wall/max(1, st['cont_steps'] - st0['cont_steps']):.0f} us, mean occupancy {(st['cont_slot_steps'] - st0['cont_slot_steps'])/max(1, st['cont_steps'] - st0['cont_steps']):.1f}" if continuous else ""), flush=True)
    disp.close()
for v in voices:
    v.close()
for m in reversed(models):
    m.close()
