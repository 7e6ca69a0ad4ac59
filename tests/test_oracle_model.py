"""CPU checks of the oracle at MODEL level (no GPU): invariants the reference states between its own code paths.

a21b -- the stateless full-sequence forward (FlowLM.FlowMain, flow_lm.go:192-233: concat text || latents, causal mask, last
token) must equal what the serving path computes statefully (PromptText + SampleNextLatentStateful, flow_lm.go:155-187,
238-299).  The reference replays this on a hand-built layer (flow_guards_test.go:366-461, transcribed in
test_oracle_kat.py); here it is replayed on a whole synthetic checkpoint, for several sequence lengths.
"""
import dataclasses
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402

import ptts_amd  # noqa: E402


@pytest.fixture(scope="module")
def tiny():
    synth = ptts_amd.load().synth
    cfg = synth.SynthConfig.tiny()
    return cfg, O.OracleModel(synth.make_checkpoint(cfg, seed=1234))


@pytest.mark.parametrize("n_text,n_steps", [(3, 1), (5, 4), (1, 7)])
def test_flow_main_equals_prompt_plus_stateful_steps(tiny, n_text, n_steps):
    """last_hidden and the EOS logit of the n-th stateful step == FlowMain over [text || BOS, frame_1 .. frame_{n-1}]."""
    cfg, om = tiny
    rng = np.random.default_rng(n_text * 10 + n_steps)
    text = om.text_embeddings(rng.integers(0, cfg.n_bins, n_text))
    st = om.new_state()
    om.prompt(st, text)
    frame = np.full(om.ldim, np.nan, np.float32)   # BOS marker (runtime_native_safetensors.go:246-253)
    seq = []
    for _ in range(n_steps):
        seq.append(frame.copy())
        frame, _, logit, last = om.step(st, frame, lsd_steps=1, eos_threshold=1e30)
    want_last, want_logit = om.flow_main(np.stack(seq), text)
    # same arithmetic in another order (full-sequence attention vs cache): the reference's own bound for this pair is 1e-5
    # on O(1) values (flow_guards_test.go:457); hidden rows here are LayerNorm outputs of O(1)
    assert np.abs(last - want_last).max() <= 2e-5, np.abs(last - want_last).max()
    assert abs(logit - want_logit) <= 2e-5 * max(1.0, abs(want_logit))
    assert st.offset(0) == n_text + n_steps


def test_flow_main_sees_the_voice_prefix_like_the_stateful_path(tiny):
    """A voice embedding is just more prompt rows (runtime_native_safetensors.go:104-119): FlowMain over [voice || text] ==
    the stateful path prompted with the same rows."""
    cfg, om = tiny
    rng = np.random.default_rng(3)
    rows = np.concatenate([rng.standard_normal((4, om.d_model)).astype(np.float32), om.text_embeddings([1, 2, 3])])
    st = om.new_state()
    om.prompt(st, rows)
    frame = np.full(om.ldim, np.nan, np.float32)
    _, _, logit, last = om.step(st, frame, eos_threshold=1e30)
    want_last, want_logit = om.flow_main(frame[None], rows)
    assert np.abs(last - want_last).max() <= 2e-5 and abs(logit - want_logit) <= 2e-5 * max(1.0, abs(want_logit))


def test_generate_reports_the_eos_logit_of_every_step(tiny):
    cfg, om = tiny
    out = om.generate([10, 20, 30], max_steps=5, eos_threshold=1e30, frames_after_eos=2)
    assert out["n_frames"] == 5 and out["eos_logits"].shape == (5,) and np.isfinite(out["eos_logits"]).all()
    # a threshold between two of the observed logits stops the loop where the first logit above it occurred (:176-192)
    lg = out["eos_logits"]
    order = np.sort(lg)
    thr = float((order[-1] + order[-2]) / 2)
    first = int(np.argmax(lg > thr))
    again = om.generate([10, 20, 30], max_steps=5, eos_threshold=thr, frames_after_eos=1)
    assert again["eos_step"] == first and again["n_frames"] == min(5, first + 2)


def test_mimi_transformer_stage_and_window_override():
    """po_mimi_transformer is the front of po_mimi_decode; a 249-key window changes exactly the rows past the window."""
    synth = ptts_amd.load().synth
    cfg = dataclasses.replace(synth.SynthConfig.tiny(), layer_scale=1.0)
    om = O.OracleModel(synth.make_checkpoint(cfg, seed=5))
    rng = np.random.default_rng(0)
    ml = om.latent_to_mimi((rng.standard_normal((17, 32)) * 0.5).astype(np.float32))
    good = om.mimi_transformer(ml)
    assert good.shape == (272, om.mimi_dim) and np.isfinite(good).all()
    om.debug_set_mimi_context(249)
    bad = om.mimi_transformer(ml)
    om.debug_set_mimi_context(250)
    assert np.array_equal(bad[:249], good[:249]) and not np.allclose(bad[249:], good[249:], atol=1e-4)
    assert np.array_equal(om.mimi_transformer(ml), good)
