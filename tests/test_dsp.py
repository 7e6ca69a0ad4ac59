"""SURVEY.md 8f N3 (optional part): PeakNormalize / DCBlock / FadeIn / FadeOut (internal/audio/dsp.go:12-78).  No GPU.
The reference's own test tables (dsp_test.go) replayed on the product, and product == oracle bit for bit for the three functions
that live in the reference's tree."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dsp as D  # noqa: E402

SR = 24000


def test_peak_normalize_table(pkg):   # dsp_test.go:8-67
    R = pkg.runtime
    for inp, want in (([0.0, 0.5, -0.25, 0.5], 1.0), ([0.1, -0.1, 0.05], 1.0), ([0.0, 1.0, -0.5], 1.0), ([0.0, 0.0, 0.0], 0.0)):
        got = R.dsp_apply(inp, normalize=True)
        assert abs(float(np.abs(got).max()) - want) <= 1e-6
        assert np.array_equal(got, D.peak_normalize(inp))
    got = R.dsp_apply([0.0, 0.25, 0.5], normalize=True)
    assert abs(got[1] / got[2] - 0.5) <= 1e-6


def test_dc_block_properties(pkg):   # dsp_test.go:69-107
    R = pkg.runtime
    got = R.dsp_apply(np.full(SR, 0.5, np.float32), dc_block=True)
    assert abs(float(got.mean())) <= 0.01
    sine = np.sin(2 * np.pi * 1000 * np.arange(SR) / SR).astype(np.float32)
    got = R.dsp_apply(sine, dc_block=True)
    assert abs(float(np.sqrt((got.astype(np.float64) ** 2).mean()) / np.sqrt((sine.astype(np.float64) ** 2).mean())) - 1.0) <= 0.01


def test_fades_table_and_bits(pkg):   # dsp_test.go:109-200
    R = pkg.runtime
    ones = np.ones(SR, np.float32)
    fi = R.dsp_apply(ones, fade_in_ms=10)
    assert fi[0] == 0.0 and fi[240] == 1.0
    fi50 = R.dsp_apply(ones, fade_in_ms=50)
    assert np.all(np.diff(fi50[:1200]) >= 0)
    fo = R.dsp_apply(ones, fade_out_ms=10)
    assert fo[-1] == 0.0 and fo[SR - 240 - 1] == 1.0
    fo50 = R.dsp_apply(ones, fade_out_ms=50)
    assert np.all(np.diff(fo50[SR - 1200:]) <= 0)
    rng = np.random.default_rng(0)
    x = rng.standard_normal(5000).astype(np.float32)
    assert np.array_equal(R.dsp_apply(x, fade_in_ms=33.3), D.fade_in(x, SR, 33.3))
    assert np.array_equal(R.dsp_apply(x, fade_out_ms=47.1), D.fade_out(x, SR, 47.1))
    assert np.array_equal(R.dsp_apply(x[:100], fade_in_ms=50, fade_out_ms=50), D.fade_out(D.fade_in(x[:100], SR, 50), SR, 50))   # fade longer than the signal
    both = R.dsp_apply(x, normalize=True, fade_in_ms=10, fade_out_ms=20)   # the CLI's order (synth.go:361-390)
    assert np.array_equal(both, D.fade_out(D.fade_in(D.peak_normalize(x), SR, 10), SR, 20))
    assert R.dsp_apply([], normalize=True, dc_block=True, fade_in_ms=5, fade_out_ms=5).size == 0
