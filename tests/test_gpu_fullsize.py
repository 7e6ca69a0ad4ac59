"""GPU parity at the reference's full shapes (`tts_b6369a24`: 6 x 1024-wide layers, ffn 4096, flow net 512 x 6,
Mimi 512 + SEANet 512/256/128/64) -- the shapes bench.py measures, on a synthetic checkpoint with the reference's tensor names.

BASELINE.json configs[1] (batch 1, f32, 63 frames) and configs[2] (batch 64, bf16 weights + KV, 125 frames, graph replay) are
compared with the oracle over their FULL length, frame by frame (the oracle needs a few seconds for a 63-frame utterance); the
error curves go to gpurun_out/parity_curve_*.json.

What "full length" can mean on random-init weights.  The AR loop feeds each frame back in, and on a random-init checkpoint that
map is expanding: ANY rounding difference grows ~1.15x per step.  The reference itself shows it -- its AVX2 build and its
scalar build (`just bench-stageprof-asm|noavx`; dot_amd64.s vs the generic loop: the oracle restates both summation orders)
start 2.6e-6 apart and are 1.5e-2 apart after 63 frames (measured here, test below).  So the frames of a long FREE-RUNNING
utterance are not a function any two correct implementations agree on, and the per-frame check is made TEACHER-FORCED instead:
at every step the GPU path is handed the oracle's previous frame (through the staged entry points = the native.Model methods),
so each of the 63 / 125 steps -- growing cache, bf16 keys and values, every kernel of the step -- is compared at the same
operating point, and the tolerance stays a single-step tolerance.  The free-running run is then held to the envelope the
reference's own two summation orders span, to bit-identity between graph replay and plain launches, and to exact EOS step /
frame count for a threshold the oracle's logits clear with a margin.
The 64-utterance batch is also covered through properties that do not depend on size: the same request in two slots gives
the same bits, whatever else shares the batch.
"""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from _parity import parity as _parity, record

pytestmark = pytest.mark.gpu

# Tolerances are set from the errors observed on MI355X (profiles/r2_parity_observed.json), at most 10x above them.
# teacher-forced steps: max |frame error| / max|frames| and |logit error| at EVERY step of the run
TF_F32_FRAME, TF_F32_LOGIT = 1.6e-4, 1.3e-4      # observed 1.6e-5 / 1.3e-5 (the reference's flow-step budget is 2e-4, python_parity_test.go:86)
TF_BF16_FRAME, TF_BF16_LOGIT = 2.7e-3, 1.8e-3    # observed 2.7e-4 / 1.8e-4; bf16 keys / values (8-bit mantissa), not a reference mode
# free-running head of an utterance (3 / 6 frames)
HEAD_F32_LAT, HEAD_F32_PCM = (2.5e-4, 6e-3), (2.4e-4, None)     # observed 2.5e-5 / 6.3e-4, 2.4e-5
HEAD_BF16_LAT, HEAD_BF16_PCM = (1.4e-2, None), (1e-2, None)     # observed 1.4e-3, 1.1e-3
# free-running, whole length.  A random-init model amplifies ANY perturbation of a frame by a fixed rate per step (the reference's own AVX2 and scalar
# builds drift apart at that rate), so the free-running error curve is held to the two quantities that are properties of the kernels rather than of
# a chaotic trajectory: where it STARTS (frame 0: the kernels' own error, <= 10x observed) and how fast it GROWS (the fitted per-step factor must be
# the model's own -- the one the reference's two summation orders show -- within FREE_RATE_SLACK: a kernel that injects error at every step would
# start low and grow faster).  Round 3 bounded the ratio of the two curves by a factor 70 around an observed 27-35: a ratio of two chaotic
# curves, which one ulp of summation order moved by 30 %; it is still recorded, no longer asserted.
FREE_F32_FRAME0 = 3.4e-4      # observed 3.4e-5 of scale (bf16 hi/lo operand split: ~13x a pure summation-order change, 2.6e-6)
FREE_RATE_SLACK = 1.05        # observed: GPU 1.158 per step, reference AVX2-vs-scalar 1.150 (ratio 1.007)


def growth_rate(curve, scale):
    """Per-step factor of an error curve: least-squares slope of log(error) over the steps before it saturates (error < 5 % of scale)."""
    c = np.asarray(curve, np.float64)
    n = int(np.argmax(c >= 0.05 * scale)) if (c >= 0.05 * scale).any() else len(c)
    n = max(n, 8)
    t = np.arange(n)
    ok = c[:n] > 0
    slope = np.polyfit(t[ok], np.log(c[:n][ok]), 1)[0]
    return float(np.exp(slope))


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parity(name, got, want, abs_tol, rel_tol=None):
    return _parity(name, got, want, (abs_tol, rel_tol))


def error_curve(name, got, ref, spf=1920):
    """Per-frame max |error| of latents and samples (absolute; the scales ride along), written next to the other GPU evidence."""
    n = ref["n_frames"]
    lat = np.abs(np.asarray(got.latents, np.float64) - ref["latents"]).max(axis=1)
    pcm_ref = np.asarray(ref["pcm"], np.float64)
    pcm_got = np.asarray(got.pcm, np.float64)
    if got.pcm.dtype == np.int16:
        pcm_got = pcm_got / 32767.0
    pcm = np.abs(pcm_got - pcm_ref).reshape(n, spf).max(axis=1)
    out = {"name": name, "frames": n, "latent_scale": float(np.abs(ref["latents"]).max()), "pcm_scale": float(np.abs(pcm_ref).max()),
           "latent_max_abs_per_frame": [float(x) for x in lat], "pcm_max_abs_per_frame": [float(x) for x in pcm]}
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"parity_curve_{name}.json"), "w") as f:
            json.dump(out, f)
    except OSError:
        pass
    print(f"[curve] {name}: latent err first/mid/last = {lat[0]:.2e} / {lat[n // 2]:.2e} / {lat[-1]:.2e} (scale {out['latent_scale']:.2f}); "
          f"pcm err = {pcm[0]:.2e} / {pcm[n // 2]:.2e} / {pcm[-1]:.2e} (scale {out['pcm_scale']:.3f})")
    return lat, pcm


def pick_threshold(logits, lo=1, hi=24):
    """A finite EOS threshold from the oracle's own logits: the step in [lo, hi) whose logit exceeds every earlier one by the
    widest margin (EOS fires at the FIRST logit above the threshold, so the step must be a running maximum); the threshold sits
    in the middle of that gap.  Early steps only: the free-running trajectories still agree there.  Returns (threshold, step, margin)."""
    best = None
    for s in range(lo, min(hi, len(logits) - 4)):
        m = float(logits[s] - logits[:s].max())
        if best is None or m > best[2]:
            best = ((float(logits[s]) + float(logits[:s].max())) / 2, s, m)
    return best


def teacher_forced(pkg, gm, om, toks_by_slot, voice, refs, steps, n_slots, cap):
    """Steps the GPU batch through `steps` frames; slots listed in `refs` get the oracle's previous frame as input at every step
    (BOS at step 0), the others feed themselves.  Returns per reference slot the per-step max |frame error| and |logit error|."""
    b = gm.new_batch(n_slots, cap)
    vs = pkg.VoiceModelState(voice)
    for sl in range(n_slots):
        b.set_voice_state(sl, vs)
    b.prompt([gm.text_embeddings(toks_by_slot[sl]) for sl in range(n_slots)])
    frames = np.full((n_slots, 32), np.nan, np.float32)
    err = {sl: ([], []) for sl in refs}
    for t in range(steps):
        for sl, ref in refs.items():
            if t > 0:
                frames[sl] = ref["latents"][t - 1]
        out, logit, _ = b.step(frames)
        for sl, ref in refs.items():
            err[sl][0].append(float(np.abs(out[sl].astype(np.float64) - ref["latents"][t]).max()))
            err[sl][1].append(float(abs(float(logit[sl]) - float(ref["eos_logits"][t]))))
        frames = out.copy()
    assert list(b.offsets()) == [125 + len(toks_by_slot[sl]) + steps for sl in range(n_slots)]
    b.close()
    return err



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
    parity("fullsize head latents (3 frames, f32)", got.latents, ref["latents"], *HEAD_F32_LAT)
    parity("fullsize head pcm (3 frames, f32)", got.pcm, ref["pcm"], *HEAD_F32_PCM)
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
    parity("fullsize head latents[0] (6 frames, bf16 weights + cache)", out[0].latents, ref["latents"], *HEAD_BF16_LAT)
    parity("fullsize head pcm[0] (6 frames, bf16 weights + cache)", out[0].pcm, ref["pcm"], *HEAD_BF16_PCM)
    dv.close()
    gm.close()
    om.close()


def test_config1_full_length_63_frames_f32(pkg, full):
    """BASELINE configs[1]: batch 1, f32 weights and cache, 25 tokens on the 125-frame voice state, 63 frames (5.04 s)."""
    cfg, paths, voice = full
    om = O.OracleModel.from_file(paths["F32"])
    gm = pkg.Model.open(paths["F32"], device=0, weights=pkg.WEIGHTS_F32, kv=pkg.KV_F32, max_batch=4)
    toks = pkg.synth.make_prompts(1, 25, 4000, seed=42)[0]
    O.set_use_avx2(True)
    ref = om.generate(toks, max_steps=63, eos_threshold=1e30, frames_after_eos=3, voice_state=voice)
    O.set_use_avx2(False)   # the reference's scalar build: same arithmetic, another summation order
    ref_scalar = om.generate(toks, max_steps=63, eos_threshold=1e30, frames_after_eos=3, voice_state=voice)
    O.set_use_avx2(True)
    scale = float(np.abs(ref["latents"]).max())
    # 1. teacher-forced: all 63 steps at the oracle's operating point
    err = teacher_forced(pkg, gm, om, [toks], voice, {0: ref}, 63, 1, 125 + 25 + 63)[0]
    fe, le = max(err[0]), max(err[1])
    print(f"[tf] config1 f32: max frame err {fe:.2e} (scale {scale:.2f}; first / last step {err[0][0]:.2e} / {err[0][-1]:.2e}), max logit err {le:.2e}")
    record("config1 teacher-forced frames (63 steps, f32)", fe, 0.0, scale, (TF_F32_FRAME, 0))
    record("config1 teacher-forced eos logits (63 steps, f32)", le, 0.0, float(np.abs(ref["eos_logits"]).max()), (TF_F32_LOGIT, 0))
    assert fe <= TF_F32_FRAME * max(1.0, scale) and le <= TF_F32_LOGIT * max(1.0, float(np.abs(ref["eos_logits"]).max())), (fe, le)
    # 2. free-running: inside the envelope of the reference's own two summation orders; graph replay == plain launches
    c = pkg.RuntimeGenerateConfig(eos_threshold=float("inf"), max_steps=63, want_latents=True, voice_model_state=pkg.VoiceModelState(voice))
    got = pkg.Runtime(gm).generate(toks, c)
    assert got.n_frames == ref["n_frames"] == 63
    lat, _ = error_curve("config1_f32_63", got, ref)
    own = np.abs(ref["latents"].astype(np.float64) - ref_scalar["latents"]).max(axis=1)
    ratio = float((lat / np.maximum.accumulate(own)).max())
    r_gpu, r_ref = growth_rate(lat, scale), growth_rate(own, scale)
    print(f"[free] reference AVX2 vs scalar order: {own[0]:.2e} -> {own[-1]:.2e} (x{r_ref:.3f} per step); GPU vs oracle: {lat[0]:.2e} -> {lat[-1]:.2e} "
          f"(x{r_gpu:.3f} per step); max ratio of the curves {ratio:.1f}")
    record("config1 free-running: frame-0 error (63 frames, f32)", float(lat[0]), 0.0, scale, (FREE_F32_FRAME0, 0))
    record("config1 free-running: per-step growth of the GPU error / of the reference's own AVX2-vs-scalar difference", r_gpu / r_ref, 0.0, 1.0, (FREE_RATE_SLACK, 0))
    record("config1 free-running: GPU error / reference's own AVX2-vs-scalar envelope (63 frames; recorded, not asserted)", ratio, 0.0, 1.0, (0, 0))
    assert lat[0] <= FREE_F32_FRAME0 * max(1.0, scale), (float(lat[0]), scale)
    assert r_gpu <= FREE_RATE_SLACK * r_ref, (r_gpu, r_ref)
    gm.set_use_graph(True)
    again = pkg.Runtime(gm).generate(toks, c)
    assert np.array_equal(again.latents, got.latents) and np.array_equal(again.pcm, got.pcm)
    # 3. EOS step and frame count under a finite threshold the oracle's logits clear with a margin far above the logit error
    thr, s_eos, margin = pick_threshold(ref["eos_logits"])
    record("config1 eos margin (oracle logits)", margin, 0.0, float(np.abs(ref["eos_logits"]).max()), (0, 0))
    assert margin > 20 * le, (margin, le)
    ref_eos = om.generate(toks, max_steps=63, eos_threshold=thr, frames_after_eos=3, voice_state=voice)
    assert ref_eos["eos_step"] == s_eos and ref_eos["n_frames"] == s_eos + 4
    for graph in (False, True):
        gm.set_use_graph(graph)
        ge = pkg.Runtime(gm).generate(toks, pkg.RuntimeGenerateConfig(eos_threshold=thr, max_steps=63, frames_after_eos=3, want_latents=True,
                                                                       voice_model_state=pkg.VoiceModelState(voice)))
        assert (ge.eos_step, ge.n_frames) == (ref_eos["eos_step"], ref_eos["n_frames"]), (graph, ge.eos_step, ge.n_frames, margin)
        assert ge.pcm.shape == ref_eos["pcm"].shape
    gm.close()
    om.close()


def test_config2_full_length_125_frames_bf16(pkg, full):
    """BASELINE configs[2]: 64 utterances, bf16 weights AND bf16 KV cache, 125 frames; graph replay (five steps per replay)
    for the free-running part.  Slots 0 and 37 against the oracle run on the same BF16 file (the oracle keeps an f32 cache: the
    bf16 cache's 8-bit mantissa on keys and values is the error measured here)."""
    cfg, paths, voice = full
    om = O.OracleModel.from_file(paths["BF16"])
    gm = pkg.Model.open(paths["BF16"], device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=64, use_graph=True)
    toks = pkg.synth.make_prompts(64, 25, 4000, seed=42)
    refs = {sl: om.generate(toks[sl], max_steps=125, eos_threshold=1e30, frames_after_eos=3, voice_state=voice) for sl in (0, 37)}
    # 1. teacher-forced: all 125 steps of slots 0 and 37 at the oracle's operating point, inside the full batch of 64
    err = teacher_forced(pkg, gm, om, list(toks), voice, refs, 125, 64, 125 + 25 + 125)
    le_max = 0.0
    for sl, ref in refs.items():
        scale = float(np.abs(ref["latents"]).max())
        fe, le = max(err[sl][0]), max(err[sl][1])
        le_max = max(le_max, le)
        print(f"[tf] config2 bf16 slot {sl}: max frame err {fe:.2e} (scale {scale:.2f}; first / last step {err[sl][0][0]:.2e} / {err[sl][0][-1]:.2e}), max logit err {le:.2e}")
        record(f"config2 teacher-forced frames[{sl}] (125 steps, bf16 weights + KV, batch 64)", fe, 0.0, scale, (TF_BF16_FRAME, 0))
        record(f"config2 teacher-forced eos logits[{sl}] (125 steps, bf16 weights + KV, batch 64)", le, 0.0, float(np.abs(ref["eos_logits"]).max()), (TF_BF16_LOGIT, 0))
        assert fe <= TF_BF16_FRAME * max(1.0, scale) and le <= TF_BF16_LOGIT * max(1.0, float(np.abs(ref["eos_logits"]).max())), (sl, fe, le)
    # 2. free-running at full length, graph replay: curve on record; bit-identical to plain launches
    dv = gm.upload_voice(pkg.VoiceModelState(voice))
    c = pkg.RuntimeGenerateConfig(eos_threshold=float("inf"), max_steps=125, want_latents=True, device_voice=dv)
    out = gm.generate_batch(list(toks), [c] * 64)
    assert all(o.n_frames == 125 for o in out)
    for sl, ref in refs.items():
        error_curve(f"config2_bf16_125_slot{sl}", out[sl], ref)
    gm.set_use_graph(False)
    plain = gm.generate_batch(list(toks), [c] * 64)
    for sl in (0, 37, 63):
        assert np.array_equal(plain[sl].latents, out[sl].latents) and np.array_equal(plain[sl].pcm, out[sl].pcm), sl
    gm.set_use_graph(True)
    # 3. finite EOS threshold per slot (each utterance stops on its own; the batch runs on until the last one has)
    thr0, s0, m0 = pick_threshold(refs[0]["eos_logits"])
    thr1, s1, m1 = pick_threshold(refs[37]["eos_logits"])
    record("config2 eos margin slot 0 (oracle logits)", m0, 0.0, float(np.abs(refs[0]["eos_logits"]).max()), (0, 0))
    record("config2 eos margin slot 37 (oracle logits)", m1, 0.0, float(np.abs(refs[37]["eos_logits"]).max()), (0, 0))
    assert min(m0, m1) > 10 * le_max, (m0, m1, le_max)
    cfgs = [pkg.RuntimeGenerateConfig(eos_threshold=float("inf"), max_steps=125, device_voice=dv) for _ in range(64)]
    cfgs[0] = pkg.RuntimeGenerateConfig(eos_threshold=thr0, max_steps=125, frames_after_eos=3, device_voice=dv)
    cfgs[37] = pkg.RuntimeGenerateConfig(eos_threshold=thr1, max_steps=125, frames_after_eos=2, device_voice=dv)
    ge = gm.generate_batch(list(toks), cfgs)
    assert (ge[0].eos_step, ge[0].n_frames) == (s0, s0 + 4), (ge[0].eos_step, ge[0].n_frames, s0, m0)
    assert (ge[37].eos_step, ge[37].n_frames) == (s1, s1 + 3), (ge[37].eos_step, ge[37].n_frames, s1, m1)
    assert ge[1].n_frames == 125 and ge[1].eos_step == -1
    # the frames an utterance produced before it stopped are the frames of the unbounded run
    assert np.array_equal(ge[0].pcm, out[0].pcm[: ge[0].pcm.shape[0]])
    dv.close()
    gm.close()
    om.close()


def test_last_stage_as_one_kernel_gives_the_two_launch_samples_bit_for_bit(pkg, full):
    """k_resblock_up (transposed convolution 128 -> 64 stride 4 + residual block + final convolution in ONE kernel; mimi.go:740-788) is taken
    when a decode has at least 8 tiles per CU -- two utterances of 100 frames do, one does not -- and keeps every product's k order: the
    samples of an utterance must be the same BITS from the fused kernel (in a batch of two) and from k_gemm_wres + k_resblock (alone).
    The head of the fused batch is also held against the oracle's decoder."""
    cfg, paths, voice = full
    gm = pkg.Model.open(paths["BF16"], device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=4)
    rng = np.random.default_rng(21)
    x = (rng.standard_normal((2, 100, cfg.ldim)) * 0.5).astype(np.float32)
    pkg.runtime.launch_counts(True)
    both, _, _ = gm.decode_stages(x)
    c2 = pkg.runtime.launch_counts(False)
    assert c2.get("k_resblock_up+final", 0) == 1 and c2.get("k_resblock+final", 0) == 0, c2
    for u in (0, 1):
        pkg.runtime.launch_counts(True)
        one, _, _ = gm.decode_stages(x[u:u + 1])
        c1 = pkg.runtime.launch_counts(False)
        assert c1.get("k_resblock+final", 0) == 1 and c1.get("k_resblock_up+final", 0) == 0, c1
        assert np.array_equal(both[u], one[0]), (u, float(np.abs(both[u] - one[0]).max()))
    om = O.OracleModel.from_file(paths["BF16"])
    want = om.mimi_decode(om.latent_to_mimi(x[0]))               # ALL 100 frames of one utterance against the oracle's decoder (192 000 samples)
    parity("fused last stage pcm (100 frames, bf16 weights)", both[0], want.reshape(-1), 2e-4, 5e-2)
    om.close()
    gm.close()


def test_last_stage_kernel_decodes_frame_ranges_behind_the_ar_loop(pkg, full):
    """k_resblock_up in RANGE mode (rows t0 .. t1 of every utterance with t0 > 0; its halo rows are recomputed from the previous stage's
    rows, which persist in the workspace): 16 utterances of 48 frames streamed in 16-frame ranges (3968 tiles per range: the fused
    kernel is taken for each) against the same call decoded at once.  Latents are the same bits (streaming only changes WHEN frames are
    decoded); every sample is delivered once and in order; the audio equals the non-streamed audio to the +/-2 LSB a differently tiled
    decode is allowed (the window attention's key tiles start elsewhere)."""
    cfg, paths, voice = full
    gm = pkg.Model.open(paths["BF16"], device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=16)
    dv = gm.upload_voice(pkg.VoiceModelState(voice))
    toks = [p for p in pkg.synth.make_prompts(16, 20, 4000, seed=9)]
    frames, per = 48, 16
    base = dict(eos_threshold=float("inf"), max_steps=frames, device_voice=dv, want_latents=True, pcm16=True)
    whole = gm.generate_batch(toks, [pkg.RuntimeGenerateConfig(**base)] * 16)
    chunks = [[] for _ in toks]

    def mk(i):
        def cb(off, x):
            chunks[i].append((off, x.copy()))
        return cb

    pkg.runtime.launch_counts(True)
    got = gm.generate_batch(toks, [pkg.RuntimeGenerateConfig(pcm_callback=mk(i), stream_frames=per, **base) for i in range(16)])
    counts = pkg.runtime.launch_counts(False)
    assert counts.get("k_resblock_up+final", 0) >= 2 and counts.get("k_resblock+final", 0) == 0, counts   # (one launch per decoded range)
    for i in range(16):
        assert got[i].n_frames == frames and np.array_equal(got[i].latents, whole[i].latents)
        offs = [o for o, _ in chunks[i]]
        sizes = [c.size for _, c in chunks[i]]
        assert offs[0] == 0 and all(offs[k + 1] == offs[k] + sizes[k] for k in range(len(offs) - 1)) and offs[-1] + sizes[-1] == frames * 1920
        assert np.array_equal(np.concatenate([c for _, c in chunks[i]]), got[i].pcm)
        assert np.abs(got[i].pcm.astype(np.int32) - whole[i].pcm.astype(np.int32)).max() <= 2, i
    # ... and the RANGE-mode audio against the oracle itself: two utterances' streamed samples (all 48 frames) vs the oracle's decoder run on the
    # very latents the GPU produced, through the reference's own int16 conversion (wav_stream.go:43-54): +/-2 LSB of rounding room on top of the
    # decoder's float tolerance
    om = O.OracleModel.from_file(paths["BF16"])
    for i in (0, 11):
        want = om.mimi_decode(om.latent_to_mimi(got[i].latents)).reshape(-1)
        scale = float(np.abs(want).max())
        want16 = (np.clip(want.astype(np.float64), -1.0, 1.0) * 32767.0).astype(np.int64)   # int16(clamp(s) * 32767): truncation toward zero
        err = np.abs(got[i].pcm.astype(np.int64) - want16).max()
        record(f"range-mode pcm16 vs oracle [{i}]", float(err), 0.0, 32767.0 * scale, (2 + 2e-4 * 32767.0 * max(scale, 1.0), None))
        assert err <= 2 + 2e-4 * 32767.0 * max(scale, 1.0), (i, int(err), scale)
    om.close()
    dv.close()
    gm.close()
