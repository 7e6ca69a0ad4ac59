"""One batch-64 call cut after two AR steps: what a rocprofv3 kernel trace of it shows is setup + prefill (+ a 2-frame decode).
    rocprofv3 --kernel-trace -d gpurun_out/pf -o pf --output-format csv -- python3 tools/prefill_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bench
import ptts_amd

pkg = ptts_amd.load()
wl = bench.WORKLOADS["b64_10s_bf16"]
cfg = pkg.synth.SynthConfig.full()
path = bench.checkpoint_path(pkg, wl["file"], 0, lambda: None)
model = pkg.Model.open(path, device=0, weights=wl["weights"], kv=wl["kv"], max_batch=64)
voice = model.upload_voice(pkg.VoiceModelState(bench.voice_modules(pkg, cfg)))
prompts = [p.tolist() for p in pkg.synth.make_prompts(64, 25, 4000, seed=3)]
c = pkg.RuntimeGenerateConfig(max_steps=int(os.environ.get("PTTS_PROBE_STEPS", "2")), eos_threshold=float("inf"), frames_after_eos=3, device_voice=voice)
for _ in range(3):
    out = model.generate_batch(prompts, [c] * 64)
print("frames", out[0].n_frames)
