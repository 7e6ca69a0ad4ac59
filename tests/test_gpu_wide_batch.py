"""More than 64 rows per AR step (round 5): the step's kernels take up to 256 utterances -- row tiles of the step linear (k_skinny, blockIdx.y; from
128 rows the layer's in_proj / linear1 / linear2 as 32- or 64-row tiles of k_tall behind k_rowprep, csrc/tall.hip -- value by value in
tests/test_gpu_tall.py), one workgroup per (utterance, head) in the step attention, 12-row tiles of the flow-net cluster (k_flow_cluster), the decoder in
groups of 64 utterances through one workspace.  The reference is batch 1 (flow_lm.go:281; internal/native/flow_transformer.go:326-389 is the layer a row goes
through); its counterpart of "how many at once" is the server's worker count (internal/server/server.go:132-134).  A wider batch must not change
what a row computes: the tests are the ones tests/test_gpu_fullsize.py runs at 64 rows -- teacher-forced against the oracle over the full 125
steps, slot symmetry bit for bit, graph replay == plain launches -- at 128 rows, and the size-independent properties at 256.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from _parity import record
from test_gpu_fullsize import TF_BF16_FRAME, TF_BF16_LOGIT, teacher_forced, full  # noqa: F401  (the tolerances of the 64-row tests; `full` is a fixture)

pytestmark = pytest.mark.gpu


def test_128_rows_teacher_forced_against_the_oracle_all_125_steps(pkg, full):
    """Slots 3 and 101 of a 128-row batch (row tiles 0 and 6 of the step linear, tiles 0 and 8 of the flow-net cluster) at the oracle's operating point
    at every one of the 125 steps, bf16 weights + bf16 KV: the tolerances of the 64-row test (tests/test_gpu_fullsize.py:36-37)."""
    cfg, paths, voice = full
    om = O.OracleModel.from_file(paths["BF16"])
    gm = pkg.Model.open(paths["BF16"], device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=128)
    toks = pkg.synth.make_prompts(128, 25, 4000, seed=43)
    refs = {sl: om.generate(toks[sl], max_steps=125, eos_threshold=1e30, frames_after_eos=3, voice_state=voice) for sl in (3, 101)}
    err = teacher_forced(pkg, gm, om, list(toks), voice, refs, 125, 128, 125 + 25 + 125)
    for sl, ref in refs.items():
        scale = float(np.abs(ref["latents"]).max())
        lscale = float(np.abs(ref["eos_logits"]).max())
        fe, le = max(err[sl][0]), max(err[sl][1])
        print(f"[tf] 128 rows, slot {sl}: max frame err {fe:.2e} (scale {scale:.2f}), max logit err {le:.2e}")
        record(f"128 rows: teacher-forced frames[{sl}] (125 steps, bf16 weights + KV)", fe, 0.0, scale, (TF_BF16_FRAME, 0))
        record(f"128 rows: teacher-forced eos logits[{sl}] (125 steps, bf16 weights + KV)", le, 0.0, lscale, (TF_BF16_LOGIT, 0))
        assert fe <= TF_BF16_FRAME * max(1.0, scale) and le <= TF_BF16_LOGIT * max(1.0, lscale), (sl, fe, le)
    gm.close()
    om.close()


@pytest.mark.parametrize("rows", [128, 256])
def test_wide_batch_is_slot_symmetric_and_graph_equals_plain(pkg, full, rows):
    """Slot i and slot rows - 1 - i carry the same prompt (neighbours differ): equal BITS in latents and samples, whichever row tile, cluster tile
    and decoder group a slot falls into (the two halves of a pair sit in different decoder groups); graph replay and plain launches give the same
    bits; the utterances a 64-row engine produces for the same prompts agree to kernel-selection rounding on the first frames."""
    cfg, paths, voice = full
    gm = pkg.Model.open(paths["BF16"], device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=rows)
    dv = gm.upload_voice(pkg.VoiceModelState(voice))
    half = pkg.synth.make_prompts(rows // 2, 25, 4000, seed=6)
    toks = [half[i] if i < rows // 2 else half[rows - 1 - i] for i in range(rows)]
    c = pkg.RuntimeGenerateConfig(eos_threshold=float("inf"), max_steps=30, want_latents=True, device_voice=dv)
    out = gm.generate_batch(toks, [c] * rows)
    assert all(o.n_frames == 30 and o.pcm.shape == (30 * 1920,) for o in out)
    for i in range(rows // 2):
        assert np.array_equal(out[i].latents, out[rows - 1 - i].latents), i
        assert np.array_equal(out[i].pcm, out[rows - 1 - i].pcm), i
    assert not np.array_equal(out[0].latents, out[1].latents)
    assert all(np.isfinite(o.pcm).all() for o in out)
    gm.set_use_graph(True)
    again = gm.generate_batch(toks, [c] * rows)
    for sl in (0, 17, rows // 2, rows - 1):
        assert np.array_equal(again[sl].latents, out[sl].latents) and np.array_equal(again[sl].pcm, out[sl].pcm), sl
    gm.set_use_graph(False)
    # the same prompts through 64 rows at a time (ptts_model_set_max_batch): other kernel variants (16-column blocks below 128 blocks), same values to rounding
    gm.set_max_batch(64)
    narrow = gm.generate_batch(toks[:64], [c] * 64)
    scale = max(float(np.abs(o.latents).max()) for o in narrow)
    d0 = max(float(np.abs(narrow[i].latents[0] - out[i].latents[0]).max()) for i in range(64))
    record(f"{rows} rows vs 64 rows: frame 0 of the same prompts", d0, 0.0, scale, (1e-3, 0))
    assert d0 <= 1e-3 * max(1.0, scale), (d0, scale)
    dv.close()
    gm.close()


def test_wide_batch_eos_and_ragged_budgets(pkg, full):
    """Per-row loop state beyond 64 rows: budgets that differ row by row (and one finite EOS threshold) end each utterance on its own step; the frames
    produced before an utterance stopped are the frames of the unbounded run (runtime_native_safetensors.go:150-201 per row)."""
    cfg, paths, voice = full
    rows = 160
    gm = pkg.Model.open(paths["BF16"], device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=rows, use_graph=True)
    dv = gm.upload_voice(pkg.VoiceModelState(voice))
    toks = pkg.synth.make_prompts(rows, 25, 4000, seed=44)
    full_c = pkg.RuntimeGenerateConfig(eos_threshold=float("inf"), max_steps=24, want_latents=True, device_voice=dv)
    base = gm.generate_batch(list(toks), [full_c] * rows)
    budgets = [4 + (7 * i) % 21 for i in range(rows)]
    cfgs = [pkg.RuntimeGenerateConfig(eos_threshold=float("inf"), max_steps=budgets[i], want_latents=True, device_voice=dv) for i in range(rows)]
    got = gm.generate_batch(list(toks), cfgs)
    for i in range(rows):
        assert got[i].n_frames == budgets[i] and got[i].eos_step == -1, (i, got[i].n_frames, budgets[i])
        assert np.array_equal(got[i].latents, base[i].latents[: budgets[i]]), i
        assert np.array_equal(got[i].pcm, base[i].pcm[: budgets[i] * 1920]), i
    dv.close()
    gm.close()
