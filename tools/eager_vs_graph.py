"""A/B: the AR step replayed as a hipGraph against the same launches issued eagerly (batch 64, 125 frames).
    python tools/eager_vs_graph.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bench
import ptts_amd

pkg = ptts_amd.load()
cfg = pkg.synth.SynthConfig.full()
for name, graph in [(n, g) for n in sys.argv[1:] or ["b64_10s_bf16", "b1_5s_f32"] for g in (True, False, True, False)]:
    wl = dict(bench.WORKLOADS["b64_10s_bf16"], batch=int(name[3:])) if name.startswith("bf:") else bench.WORKLOADS[name]   # "bf:16" = the bf16 workload at batch 16
    B = wl["batch"]
    path = bench.checkpoint_path(pkg, wl["file"], 0, lambda: None)
    prompts = [p.tolist() for p in pkg.synth.make_prompts(B, 25, 4000, seed=3)]
    model = pkg.Model.open(path, device=0, weights=wl["weights"], kv=wl["kv"], max_batch=B, use_graph=graph)
    voice = model.upload_voice(pkg.VoiceModelState(bench.voice_modules(pkg, cfg)))
    c = pkg.RuntimeGenerateConfig(max_steps=wl["frames"], eos_threshold=float("inf"), frames_after_eos=3, device_voice=voice)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        out = model.generate_batch(prompts, [c] * B)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name} use_graph={graph}: ms per batch {['%.1f' % t for t in ts]}", flush=True)
    del out
    voice.close()
    model.close()
