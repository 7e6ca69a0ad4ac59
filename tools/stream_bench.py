"""Time to first audio with frame-granular streaming (pcm_callback): python tools/stream_bench.py"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bench
import ptts_amd

pkg = ptts_amd.load()
cfg = pkg.synth.SynthConfig.full()
for wl_name, batches in (("b64_10s_bf16", (1, 8, 64)), ("b1_5s_f32", (1,))):
    wl = bench.WORKLOADS[wl_name]
    path = bench.checkpoint_path(pkg, wl["file"], 0, lambda: None)
    model = pkg.Model.open(path, device=0, weights=wl["weights"], kv=wl["kv"], max_batch=64)
    voice = model.upload_voice(pkg.VoiceModelState(bench.voice_modules(pkg, cfg)))
    prompts = [p.tolist() for p in pkg.synth.make_prompts(64, 25, 4000, seed=3)]
    for B in batches:
        for sf in (0, 12, 4):
            first, total = [], []
            for it in range(5):
                t_first = [None]
                t0 = time.perf_counter()

                def cb(off, x, t_first=t_first, t0=t0):
                    if t_first[0] is None:
                        t_first[0] = time.perf_counter() - t0

                cfgs = [pkg.RuntimeGenerateConfig(max_steps=125, eos_threshold=float("inf"), frames_after_eos=3, device_voice=voice, pcm16=True,
                                                  pcm_callback=(cb if (sf and i == 0) else None), stream_frames=sf) for i in range(B)]
                model.generate_batch(prompts[:B], cfgs)
                total.append(time.perf_counter() - t0)
                first.append(t_first[0] if t_first[0] is not None else total[-1])
            print(f"{wl_name:13s} batch {B:2d}  stream_frames {sf:2d}: first audio {1e3*statistics.median(first[1:]):7.2f} ms   whole 10-s utterance {1e3*statistics.median(total[1:]):7.2f} ms", flush=True)
    voice.close()
    model.close()
