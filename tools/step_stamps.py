"""In-situ phase timeline of every step-linear launch of ONE real AR step at batch 64 / bf16 (cold caches, real operands):
    python tools/step_stamps.py
Ticks are shader-clock cycles (s_memtime); entry stamps also carry the 100 MHz wall clock, which gives each launch's start time."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ptts_amd

pkg = ptts_amd.load()
L = pkg.runtime.hooks()   # the measurement entry points live in libptts_hooks.so (include/ptts_debug.h)
wl = bench.WORKLOADS["b64_10s_bf16"]
cfg = pkg.synth.SynthConfig.full()
path = bench.checkpoint_path(pkg, wl["file"], 0, lambda: None)
model = pkg.Model.open(path, device=0, weights=wl["weights"], kv=wl["kv"], max_batch=64)
voice = pkg.VoiceModelState(bench.voice_modules(pkg, cfg))
B = int(os.environ.get("PTTS_PROBE_BATCH", "64"))
b = model.new_batch(B, 320)
for sl in range(B):
    b.set_voice_state(sl, voice)
prompts = pkg.synth.make_prompts(B, 25, 4000, seed=3)
b.prompt([model.text_embeddings(p) for p in prompts])
frames = np.full((B, 32), np.nan, np.float32)
for _ in range(4):
    frames, _, _ = b.step(frames)
cap = 20000
buf = np.zeros((cap, 8), np.uint64)
desc = np.zeros((128, 8), np.int32)
n = C.c_int32(0)
L.ptts_debug_step_stamps.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
rc = L.ptts_debug_step_stamps(b.h, 1, buf.ctypes.data, cap, desc.ctypes.data, 128, C.byref(n))
assert rc == 0, L.ptts_last_error().decode()
names = ["sums", "st.issued", "staged", "mfma", "k-red", "stored"]   # slots 1..6; 1 and 2 are stamped in the epilogue (sums final, stores issued)
off = 0
t_first = None
print(f"{'launch':28s} {'blocks':>6s} {'start us':>8s} " + " ".join(f"{x:>9s}" for x in names) + "   (p50 ticks since block entry; last column p100 stored)")
for i in range(n.value):
    M, N, K, pro, nj, cg, blocks, sk = desc[i]
    s = buf[off:off + blocks].astype(np.int64)
    off += blocks
    real = s[:, 7]
    if t_first is None:
        t_first = real.min()
    rel = s[:, 1:7] - s[:, 0:1]
    p50 = np.percentile(rel, 50, axis=0)
    print(f"{N:5d}x{K:4d} pro={pro:2d} nj={nj} cg={cg} z={sk} {blocks:6d} {(real.min() - t_first) / 100.0:8.2f} " + " ".join(f"{int(v):9d}" for v in p50) +
          f"   {int(rel[:, 5].max()):6d}  entry spread {(real.max() - real.min()) * 10} ns")
b.close()
model.close()
