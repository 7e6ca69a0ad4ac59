#!/usr/bin/env python3
"""What the f32-grade sums of the decoder cost, and what they buy: the library named by PTTS_LIB_PATH (the shipped one, or tools/probes/hi_only/build/libptts_hip.so:
every decoder product with the hi half of its activation split only) decodes two 125-frame utterances -- PCM against the ORACLE's decoder on the same latents
(oracle/ptts_oracle.c: mimi.go:719-789 in f32 with f64 accumulations) -- and the benchmark batch (64 x 125 frames), decoder phase timed on the device.
The reference's own decode tolerance is rel 5e-2 (native/python_parity_test.go:119-120)."""
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

import bench  # noqa: E402
import ptts_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    pkg = ptts_amd.load()
    cfg = pkg.synth.SynthConfig.full()
    path = bench.checkpoint_path(pkg, "BF16", 0, lambda: None)
    model = pkg.Model.open(path, device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=64, use_graph=True)
    voice_mods = bench.voice_modules(pkg, cfg)
    voice = model.upload_voice(pkg.VoiceModelState(voice_mods))
    prompts = pkg.synth.make_prompts(64, 25, 4000, seed=42)
    wl = dict(batch=64, frames=125)
    toks = [np.ascontiguousarray(p, np.int64) for p in prompts]
    c2 = pkg.RuntimeGenerateConfig(temperature=0.0, eos_threshold=float("inf"), max_steps=125, lsd_decode_steps=1, frames_after_eos=3, device_voice=voice, want_latents=True)
    two = model.generate_batch(toks[:2], [c2] * 2)
    om = O.OracleModel.from_file(path)
    worst_abs = worst_rel = scale = 0.0
    for r in two:
        ref = om.mimi_decode(om.latent_to_mimi(r.latents)).reshape(-1).astype(np.float64)
        got = r.pcm.astype(np.float64)
        err = np.abs(got - ref)
        sc = float(np.abs(ref).max())
        big = np.abs(ref) >= 1e-3 * sc
        worst_abs = max(worst_abs, float(err.max()))
        worst_rel = max(worst_rel, float((err[big] / np.abs(ref[big])).max()))
        scale = max(scale, sc)
        rms = float(np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean()))
        print(f"  utterance: max |pcm - oracle| {err.max():.3e} (scale {sc:.3f}), max rel (|ref| >= 1e-3 scale) {(err[big] / np.abs(ref[big])).max():.3e}, rms error / rms signal {rms:.3e}")
    om.close()
    cfgs = bench.gen_cfgs(pkg, wl, 64, voice)
    for _ in range(2):
        model.generate_batch(toks, cfgs)
    mimi = []
    for _ in range(5):
        model.profile_enable(2)
        try:
            model.generate_batch(toks, cfgs)
            mimi.append(model.profile_read()["mimi_ms"])
        finally:
            model.profile_enable(False)
    print(f"library {os.environ.get('PTTS_LIB_PATH', 'go-pocket-tts_amd/libptts_hip.so')}: decoder {statistics.median(mimi):.2f} ms per 64 x 125 frames (5 runs: {min(mimi):.2f}..{max(mimi):.2f}); "
          f"PCM vs oracle on 2 x 125 frames: max abs {worst_abs:.3e} of scale {scale:.3f}, max rel {worst_rel:.3e}")
    voice.close()
    model.close()


if __name__ == "__main__":
    main()
