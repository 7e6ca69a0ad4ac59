"""Temperature > 0 as a first-class path (reference default Temperature 0.7, internal/config/config.go:99): the per-step
sampling noise of FlowLM.makeGaussianNoise (flow_lm.go:386-408) is drawn on the device per (noise_seed, step); the same rows
can be read back (ptts_noise_rows) and handed to the oracle, which makes temperature-0.7 generation a parity case.
"""
import numpy as np
import pytest

from oracle import oracle as O
from _parity import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny(pkg, tmp_path_factory):
    synth = pkg.synth
    cfg = synth.SynthConfig.tiny()
    path = str(tmp_path_factory.mktemp("ckpt") / "tiny.safetensors")
    synth.write_safetensors(path, synth.make_checkpoint(cfg, seed=1234))
    om = O.OracleModel.from_file(path)
    gm = pkg.Model.open(path, device=0, max_batch=64)
    yield cfg, om, gm
    gm.close()
    om.close()


def test_device_draw_is_standard_normal_times_sqrt_temperature(tiny):
    """makeGaussianNoise: data[i] = NormFloat64() * sqrt(max(temperature, 0)).  Moments of 4096 x 32 draws (sigma^2 = 0.7):
    mean, variance, skewness, kurtosis within 5 standard errors; temperature <= 0 gives zeros (flow_lm.go:395-404)."""
    _, _, gm = tiny
    x = gm.noise_rows(12345, 0.7, 4096).astype(np.float64).ravel()
    n = x.size
    z = x / np.sqrt(0.7)
    assert abs(z.mean()) < 5 / np.sqrt(n)
    assert abs(z.var() - 1.0) < 5 * np.sqrt(2.0 / n)
    assert abs((z ** 3).mean()) < 5 * np.sqrt(15.0 / n)
    assert abs((z ** 4).mean() - 3.0) < 5 * np.sqrt(96.0 / n)
    assert np.abs(z).max() < 6.5 and (np.abs(z) > 3).mean() == pytest.approx(0.0027, abs=0.0012)
    # no lattice / repetition: all rows distinct, neighbouring elements uncorrelated
    assert len({r.tobytes() for r in gm.noise_rows(12345, 0.7, 4096)}) == 4096
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 5 / np.sqrt(n)
    assert not gm.noise_rows(1, 0.0, 8).any() and not gm.noise_rows(1, -1.0, 8).any()   # TestMakeGaussianNoise: sigma = 0 for temp < 0


def test_draw_is_a_function_of_seed_and_step_only(tiny):
    _, _, gm = tiny
    a, b = gm.noise_rows(7, 0.7, 16), gm.noise_rows(7, 0.7, 64)
    assert np.array_equal(a, b[:16])                       # row `step` does not depend on how many rows are drawn
    assert not np.array_equal(a, gm.noise_rows(8, 0.7, 16))
    assert np.allclose(gm.noise_rows(7, 2.8, 16), 2.0 * a, rtol=1e-6)   # sigma = sqrt(temperature)


def test_temperature_changes_the_audio_and_seeds_reproduce_it(pkg, tiny):
    _, _, gm = tiny
    rt = pkg.Runtime(gm)
    base = dict(eos_threshold=float("inf"), max_steps=4, want_latents=True)
    cold = rt.generate([10, 20, 30], pkg.RuntimeGenerateConfig(temperature=0.0, **base))
    a = rt.generate([10, 20, 30], pkg.RuntimeGenerateConfig(temperature=0.7, noise_seed=99, **base))
    b = rt.generate([10, 20, 30], pkg.RuntimeGenerateConfig(temperature=0.7, noise_seed=99, **base))
    c = rt.generate([10, 20, 30], pkg.RuntimeGenerateConfig(temperature=0.7, noise_seed=100, **base))
    assert np.array_equal(a.latents, b.latents) and np.array_equal(a.pcm, b.pcm)
    assert not np.array_equal(a.latents, cold.latents) and not np.array_equal(a.latents, c.latents)
    # no seed named: every request gets a fresh stream (the reference's rng is seeded with the clock)
    d = rt.generate([10, 20, 30], pkg.RuntimeGenerateConfig(temperature=0.7, **base))
    e = rt.generate([10, 20, 30], pkg.RuntimeGenerateConfig(temperature=0.7, **base))
    assert not np.array_equal(d.latents, e.latents) and not np.array_equal(d.latents, cold.latents)


@pytest.mark.parametrize("graph", [False, True])
def test_batch_of_64_at_temperature_0p7_matches_the_oracle_on_the_same_noise(pkg, tiny, graph):
    """64 utterances, temperature 0.7, device-drawn noise, one seed per request; the oracle gets each request's rows through
    its injected-noise input.  Plain launches and graph replay."""
    cfg, om, gm = tiny
    gm.set_use_graph(graph)
    toks = pkg.synth.make_prompts(64, 5, cfg.n_bins, seed=3)
    steps = 6
    cfgs = [pkg.RuntimeGenerateConfig(temperature=0.7, noise_seed=1000 + i, eos_threshold=float("inf"), max_steps=steps, want_latents=True)
            for i in range(64)]
    out = gm.generate_batch(list(toks), cfgs)
    for i in (0, 1, 17, 40, 63):
        nz = gm.noise_rows(1000 + i, 0.7, steps)
        ref = om.generate(toks[i], max_steps=steps, eos_threshold=1e30, noise=nz)
        assert out[i].n_frames == ref["n_frames"] == steps
        parity(f"T=0.7 latents[{i}] graph={graph}", out[i].latents, ref["latents"], (2e-4, 5e-3))
        parity(f"T=0.7 pcm[{i}] graph={graph}", out[i].pcm, ref["pcm"], (2e-4, 5e-2))
    gm.set_use_graph(False)


def test_injected_noise_that_is_too_short_is_refused(pkg, tiny):
    _, _, gm = tiny
    rt = pkg.Runtime(gm)
    with pytest.raises(pkg.runtime.PttsError) as e:
        rt.generate([1, 2, 3], pkg.RuntimeGenerateConfig(temperature=0.7, noise=np.zeros((3, 32), np.float32), eos_threshold=float("inf"), max_steps=5))
    assert "noise has 3 rows" in str(e.value) and "step budget is 5" in str(e.value)
    ok = rt.generate([1, 2, 3], pkg.RuntimeGenerateConfig(temperature=0.7, noise=np.zeros((5, 32), np.float32), eos_threshold=float("inf"), max_steps=5))
    assert ok.n_frames == 5
