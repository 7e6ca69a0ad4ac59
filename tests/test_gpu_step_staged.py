"""The AR step through the staged entry points (ptts_batch_prompt / ptts_batch_step), teacher-forced against the oracle on tiny
f32 and bf16 checkpoints, and the whole loop plain vs replayed from a hipGraph.  (Round 2 ran this over three launch plans; the two
that measured no faster left the library in round 3 -- tools/probes/step_plans/README.md.)"""
import numpy as np
import pytest

from oracle import oracle as O
from _parity import parity

pytestmark = pytest.mark.gpu

FLOW_TOL = (2e-4, 5e-3)


@pytest.fixture(scope="module", params=["F32", "BF16"])
def ckpt(request, pkg, tmp_path_factory):
    synth = pkg.synth
    cfg = synth.SynthConfig.tiny()
    path = str(tmp_path_factory.mktemp("ckpt") / f"tiny_{request.param}.safetensors")
    synth.write_safetensors(path, synth.make_checkpoint(cfg, seed=1234), dtype=request.param)
    om = O.OracleModel.from_file(path)
    yield cfg, path, request.param, om
    om.close()


def test_staged_step_outputs_against_the_oracle(pkg, ckpt):
    cfg, path, dtype, om = ckpt
    gm = pkg.Model.open(path, device=0, weights=1 if dtype == "BF16" else 0)
    toks = [np.array([3, 1, 4, 1, 5], np.int64), np.array([9, 2, 6], np.int64), np.array([5, 3, 5, 8, 9, 7, 9], np.int64)]
    b = gm.new_batch(3, 64)
    b.prompt([gm.text_embeddings(t) for t in toks])
    states = []
    for t in toks:
        st = om.new_state()
        om.prompt(st, om.text_embeddings(t))
        states.append(st)
    frames = np.full((3, 32), np.nan, np.float32)
    pkg.runtime.launch_counts(True)
    for step in range(4):
        out, logit, last = b.step(frames)
        want = [om.step(st, frames[i], eos_threshold=1e30) for i, st in enumerate(states)]
        for i, (w_out, _, w_logit, w_last) in enumerate(want):
            parity(f"staged {dtype} step {step} last_hidden[{i}]", last[i], w_last, FLOW_TOL)
            parity(f"staged {dtype} step {step} frame[{i}]", out[i], w_out, FLOW_TOL, rel_floor=1e-2)
            assert abs(float(logit[i]) - w_logit) <= FLOW_TOL[0] * max(1.0, abs(w_logit)), (step, i)
        frames = np.stack([w[0] for w in want])   # teacher-forced
    counts = pkg.runtime.launch_counts(False)
    assert counts.get("k_skinny", 0) > 0 and counts.get("k_attn_step", 0) > 0, counts   # the step kernels are the ones that ran
    b.close()
    gm.close()


def test_loop_end_to_end_plain_and_under_graph_replay(pkg, ckpt):
    cfg, path, dtype, om = ckpt
    toks = pkg.synth.make_prompts(9, 6, cfg.n_bins, seed=11)
    gm = pkg.Model.open(path, device=0, weights=1 if dtype == "BF16" else 0, max_batch=16)
    c = pkg.RuntimeGenerateConfig(eos_threshold=float("inf"), max_steps=5, want_latents=True)
    plain = gm.generate_batch(list(toks), [c] * 9)
    gm.set_use_graph(True)
    graph = gm.generate_batch(list(toks), [c] * 9)
    for a, g in zip(plain, graph):
        assert np.array_equal(a.latents, g.latents) and np.array_equal(a.pcm, g.pcm)
    gm.close()
    ref = om.generate(toks[4], max_steps=5, eos_threshold=1e30, frames_after_eos=3)
    parity(f"loop {dtype} latents[4] vs oracle", plain[4].latents, ref["latents"], (2.5e-4, 5e-2))
    parity(f"loop {dtype} pcm[4] vs oracle", plain[4].pcm, ref["pcm"], (3e-4, 1e-1))
