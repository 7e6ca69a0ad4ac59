"""Where does the wall time of generate_batch go: the C call or the numpy copies of the results?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import ptts_amd
import bench
pkg = ptts_amd.load()
wl = bench.WORKLOADS["b64_10s_bf16"]
cfg = pkg.synth.SynthConfig.full()
path = bench.checkpoint_path(pkg, "BF16", 0, lambda: None)
model = pkg.Model.open(path, device=0, weights=1, kv=1)
voice = model.upload_voice(pkg.VoiceModelState(bench.voice_modules(pkg, cfg)))
prompts = [p.tolist() for p in pkg.synth.make_prompts(64, 25, 4000, seed=1)]
cfgs = [pkg.RuntimeGenerateConfig(max_steps=125, eos_threshold=1e30, lsd_decode_steps=1, frames_after_eos=3, device_voice=voice) for _ in prompts]
import ctypes as C
R = pkg.runtime
tprev = time.perf_counter()
for it in range(6):
    t0 = time.perf_counter()
    out = model.generate_batch(prompts, cfgs)
    t1 = time.perf_counter()
    print(f"iter {it}: gap before call {1e3*(t0-tprev):.1f} ms, generate_batch {1e3*(t1-t0):.1f} ms")
    tprev = t1
a = np.zeros(240000 * 64, np.float32)
t0 = time.perf_counter(); b = a.copy(); t1 = time.perf_counter()
print(f"plain numpy copy of 61 MB: {1e3*(t1-t0):.1f} ms")
