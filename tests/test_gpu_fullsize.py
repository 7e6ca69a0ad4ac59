"""GPU parity at the reference's full shapes (`tts_b6369a24`: 6 x 1024-wide layers, ffn 4096, flow net 512 x 6,
Mimi 512 + SEANet 512/256/128/64) -- the shapes bench.py measures, on a synthetic checkpoint with the reference's tensor names.

The oracle finishes a few frames of the full model in seconds, so the head of an utterance is compared against it directly;
the full batch is covered through properties that do not depend on size: the same request in two slots of a 64-utterance
batch gives the same bits, whatever else shares the batch.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def parity(name, got, want, abs_tol, rel_tol=None):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape and np.isfinite(got).all(), name
    err = np.abs(got - want)
    scale = max(1.0, np.abs(want).max())
    assert err.max() <= abs_tol * scale, f"{name}: max abs {err.max():.3e} (scale {scale:.2f})"
    if rel_tol is not None:
        big = np.abs(want) >= 1e-3 * np.abs(want).max()
        assert (err[big] / np.abs(want[big])).max() <= rel_tol, name


@pytest.fixture(scope="module")
def full(pkg):
    import bench
    cfg = pkg.synth.SynthConfig.full()
    paths = {d: bench.checkpoint_path(pkg, d, 0, lambda: None) for d in ("F32", "BF16")}
    return cfg, paths, bench.voice_modules(pkg, cfg)


def test_full_size_head_of_an_utterance_against_the_oracle(pkg, full):
    """f32 weights, f32 cache (BASELINE configs[1]): 12 tokens on the 125-frame voice state, 3 frames -- prefill, three AR steps
    through all 6 layers and the flow net, and the whole decoder, against the CPU restatement (multi-step tolerances of
    test_gpu_model.py: accumulated rounding over autoregressive steps)."""
    cfg, paths, voice = full
    om = O.OracleModel.from_file(paths["F32"])
    gm = pkg.Model.open(paths["F32"], device=0, weights=pkg.WEIGHTS_F32, kv=pkg.KV_F32, max_batch=4)
    toks = pkg.synth.make_prompts(1, 12, 4000, seed=5)[0]
    ref = om.generate(toks, max_steps=3, eos_threshold=1e30, frames_after_eos=3, voice_state=voice)
    got = pkg.Runtime(gm).generate(toks, pkg.RuntimeGenerateConfig(eos_threshold=float("inf"), max_steps=3, want_latents=True,
                                                                    voice_model_state=pkg.VoiceModelState(voice)))
    assert got.n_frames == ref["n_frames"] == 3
    parity("latents", got.latents, ref["latents"], 2e-3, 5e-2)
    parity("pcm", got.pcm, ref["pcm"], 5e-3)
    gm.close()
    om.close()


def test_full_size_batch_of_64_is_slot_independent_and_tracks_the_oracle(pkg, full):
    """bf16 weights and cache, 64 utterances (BASELINE configs[2]) for 6 frames: slot i and slot 63 - i carry the same prompt
    (neighbours differ), so their latents and samples must be identical bit for bit -- every kernel of the path treats a
    batch row independently of its position and of the other rows; slot 0 is also held against the oracle on the same BF16
    file (bf16 cache: 8-bit mantissa keys/values, max-norm bound as in test_gpu_model.py)."""
    cfg, paths, voice = full
    gm = pkg.Model.open(paths["BF16"], device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=64)
    dv = gm.upload_voice(pkg.VoiceModelState(voice))
    half = pkg.synth.make_prompts(32, 25, 4000, seed=6)
    toks = [half[i] if i < 32 else half[63 - i] for i in range(64)]
    c = pkg.RuntimeGenerateConfig(eos_threshold=float("inf"), max_steps=6, want_latents=True, device_voice=dv)
    out = gm.generate_batch(toks, [c] * 64)
    assert all(o.n_frames == 6 and o.pcm.shape == (6 * 1920,) for o in out)
    for i in range(32):
        assert np.array_equal(out[i].latents, out[63 - i].latents), i
        assert np.array_equal(out[i].pcm, out[63 - i].pcm), i
    assert not np.array_equal(out[0].latents, out[1].latents)
    # the benchmark's own shape: 125 frames (10 s) per utterance, int16 results written by the decoder's last kernel
    c125 = pkg.RuntimeGenerateConfig(eos_threshold=float("inf"), max_steps=125, device_voice=dv, pcm16=True)
    long = gm.generate_batch(toks, [c125] * 64)
    assert all(o.n_frames == 125 and o.pcm.shape == (125 * 1920,) and o.pcm.dtype == np.int16 for o in long)
    for i in range(32):
        assert np.array_equal(long[i].pcm, long[63 - i].pcm), i
    # causal: the first frames do not depend on how long the run is (decoding 6 or 125 frames may pick other GEMM tilings: 1 LSB)
    head = long[0].pcm[: 6 * 1920].astype(np.int32) - O.pcm16(out[0].pcm).astype(np.int32)
    assert np.abs(head).max() <= 1
    om = O.OracleModel.from_file(paths["BF16"])
    ref = om.generate(toks[0], max_steps=6, eos_threshold=1e30, frames_after_eos=3, voice_state=voice)
    parity("latents[0] (bf16 cache)", out[0].latents, ref["latents"], 3e-2)
    parity("pcm[0] (bf16 cache)", out[0].pcm, ref["pcm"], 3e-2)
    dv.close()
    gm.close()
    om.close()
