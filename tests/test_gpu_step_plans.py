"""The three launch plans of the AR step (ptts_opts.step_plan) compute the same step: 0 = default (f32 edges, LayerNorms in the
consumers' prologues), 1 = split planes on the flow net's mlp0 -> mlp2 edge, 2 = planes on every edge with norm1 / out_norm as
their own launch and norm2 folded into linear1's epilogue (rstd * ((x*g) W^T - mean * wg) + wb).  Each against the oracle, staged (prefill K/V, step
outputs) and end to end, f32 and bf16 weights; plan 2 is asserted to run the kernels it is about."""
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


@pytest.mark.parametrize("plan", [0, 1, 2])
def test_step_outputs_of_every_plan_against_the_oracle(pkg, ckpt, plan):
    cfg, path, dtype, om = ckpt
    gm = pkg.Model.open(path, device=0, weights=1 if dtype == "BF16" else 0, step_plan=plan)
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
            parity(f"plan {plan} {dtype} step {step} last_hidden[{i}]", last[i], w_last, FLOW_TOL)
            parity(f"plan {plan} {dtype} step {step} frame[{i}]", out[i], w_out, FLOW_TOL, rel_floor=1e-2)
            assert abs(float(logit[i]) - w_logit) <= FLOW_TOL[0] * max(1.0, abs(w_logit)), (plan, step, i)
        frames = np.stack([w[0] for w in want])   # teacher-forced
    counts = pkg.runtime.launch_counts(False)
    assert ("k_combine_ln" in counts) == (plan == 2), counts
    b.close()
    gm.close()


def test_plans_agree_end_to_end_and_under_graph_replay(pkg, ckpt):
    cfg, path, dtype, om = ckpt
    toks = pkg.synth.make_prompts(9, 6, cfg.n_bins, seed=11)
    outs = {}
    for plan in (0, 1, 2):
        gm = pkg.Model.open(path, device=0, weights=1 if dtype == "BF16" else 0, step_plan=plan, max_batch=16)
        c = pkg.RuntimeGenerateConfig(eos_threshold=float("inf"), max_steps=5, want_latents=True)
        plain = gm.generate_batch(list(toks), [c] * 9)
        gm.set_use_graph(True)
        graph = gm.generate_batch(list(toks), [c] * 9)
        for a, g in zip(plain, graph):
            assert np.array_equal(a.latents, g.latents) and np.array_equal(a.pcm, g.pcm), plan
        outs[plan] = plain
        gm.close()
    ref = om.generate(toks[4], max_steps=5, eos_threshold=1e30, frames_after_eos=3)
    for plan in (0, 1, 2):
        parity(f"plan {plan} {dtype} latents[4] vs oracle", outs[plan][4].latents, ref["latents"], (2.5e-4, 5e-2))
        parity(f"plan {plan} {dtype} pcm[4] vs oracle", outs[plan][4].pcm, ref["pcm"], (3e-4, 1e-1))
