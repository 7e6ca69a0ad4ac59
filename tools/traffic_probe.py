"""Short batch-64 bf16 run (prefill + PTTS_PROBE_STEPS AR steps + their decode, PTTS_PROBE_REPS = 3 times over; plain launches, or graph replay with
PTTS_PROBE_GRAPH=1): what bench.py's rocprofv3 counter passes profile for roofline.traffic, and its kernel-trace pass for avg_launch_us_rocprof.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -o pmc -- python3 tools/traffic_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ptts_amd

pkg = ptts_amd.load()
wl = bench.WORKLOADS["b64_10s_bf16"]
cfg = pkg.synth.SynthConfig.full()
path = bench.checkpoint_path(pkg, wl["file"], 0, lambda: None)
model = pkg.Model.open(path, device=0, weights=wl["weights"], kv=wl["kv"], max_batch=64, use_graph=os.environ.get("PTTS_PROBE_GRAPH", "0") == "1")
voice = model.upload_voice(pkg.VoiceModelState(bench.voice_modules(pkg, cfg)))
prompts = [p.tolist() for p in pkg.synth.make_prompts(64, 25, 4000, seed=3)]
c = pkg.RuntimeGenerateConfig(max_steps=int(os.environ.get("PTTS_PROBE_STEPS", "12")), eos_threshold=float("inf"), frames_after_eos=3, device_voice=voice)
for _ in range(int(os.environ.get("PTTS_PROBE_REPS", "3"))):
    out = model.generate_batch(prompts, [c] * 64)
print("frames", out[0].n_frames)
voice.close()
model.close()
