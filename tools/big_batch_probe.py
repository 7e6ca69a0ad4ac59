#!/usr/bin/env python3
"""One-shot batches beyond 64 rows (round 5): 64 / 128 / 256 utterances x 125 frames through ONE engine, graph replay, with the device-time phases.
Prints per batch size: ms per pass, x real time, prefill / AR loop / Mimi ms, us per AR step and the step-level HBM fraction
(170.5 MB of bf16 weights + the KV bytes the step reads, over 8 TB/s).  PTTS_PROBE_BATCHES=64,128 picks the sizes."""
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: F401  (one HIP runtime for the process: go-pocket-tts_amd/runtime.py lib())

import bench
import ptts_amd


def main():
    pkg = ptts_amd.load()
    cfg = pkg.synth.SynthConfig.full()
    path = bench.checkpoint_path(pkg, "BF16", 0, lambda: None)
    sizes = [int(x) for x in os.environ.get("PTTS_PROBE_BATCHES", "64,128,256").split(",")]
    frames = int(os.environ.get("PTTS_PROBE_FRAMES", "125"))
    reps = int(os.environ.get("PTTS_PROBE_REPS", "4"))
    model = pkg.Model.open(path, device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=max(sizes), use_graph=os.environ.get("PTTS_PROBE_GRAPH", "1") != "0")
    voice = model.upload_voice(pkg.VoiceModelState(bench.voice_modules(pkg, cfg)))
    prompts = pkg.synth.make_prompts(max(sizes), 25, 4000, seed=42)
    for B in sizes:
        wl = dict(batch=B, frames=frames)
        cfgs = bench.gen_cfgs(pkg, wl, B, voice)
        toks = [np.ascontiguousarray(p, np.int64) for p in prompts[:B]]
        for _ in range(2):
            out = model.generate_batch(toks, cfgs)
        torch.cuda.synchronize()
        lat = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = model.generate_batch(toks, cfgs)
            lat.append(time.perf_counter() - t0)
        assert all(o.n_frames == frames for o in out)
        ms = 1e3 * statistics.median(lat)
        model.profile_enable(2)
        try:
            model.generate_batch(toks, cfgs)
            ph = model.profile_read()
        finally:
            model.profile_enable(False)
        step_us = 1e3 * ph["ar_loop_ms"] / frames
        print(f"B={B:4d}: {ms:8.2f} ms per pass = {B * frames * 0.08 / (ms * 1e-3):9.0f} x real time | prefill {ph['prefill_ms']:.2f} ar_loop {ph['ar_loop_ms']:.2f} "
              f"mimi {ph['mimi_ms']:.2f} ms | {step_us:.1f} us per step | step-level HBM fraction {bench.step_level_fraction(B, frames, step_us):.3f}", flush=True)
    voice.close()
    model.close()


if __name__ == "__main__":
    main()
