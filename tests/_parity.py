"""Shared parity metric of the GPU tests, with a log of what was OBSERVED.

Metric = the reference's (internal/native/parity.go:20-70): max abs error and max relative error, the relative error
evaluated on elements with |want| >= 1e-3 * max|want| (a relative error on a value that is itself rounding noise says
nothing).  Every call appends {name, max_abs, max_rel, scale, tol} to gpurun_out/parity_observed.jsonl (when that
directory can be written), so that tolerances can be set from measurements: the bar is <= 10x the observed error
(profiles/r2_parity_observed.json is the committed copy of one GPU run).
"""
import json
import os

import numpy as np

_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_observed.jsonl")


def observe(name, got, want, rel_floor=1e-3):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert np.isfinite(got).all(), f"{name}: non-finite output"
    err = np.abs(got - want)
    amax = float(np.abs(want).max()) if want.size else 0.0
    big = np.abs(want) >= rel_floor * amax
    rel = float((err[big] / np.abs(want[big])).max()) if big.any() and amax > 0 else 0.0
    return float(err.max()) if err.size else 0.0, rel, amax


def record(name, max_abs, max_rel, scale, tol):
    try:
        os.makedirs(os.path.dirname(_LOG), exist_ok=True)
        with open(_LOG, "a") as f:
            f.write(json.dumps({"name": name, "max_abs": max_abs, "max_rel": max_rel, "scale": scale, "tol": list(tol)}) + "\n")
    except OSError:
        pass


def parity(name, got, want, tol, scale_abs=True, rel_floor=1e-3):
    """tol = (abs, rel); abs is multiplied by max(1, max|want|) when scale_abs; rel may be None and is evaluated on the
    elements with |want| >= rel_floor * max|want|."""
    max_abs, max_rel, amax = observe(name, got, want, rel_floor)
    ref = max(1.0, amax) if scale_abs else 1.0
    record(name, max_abs, max_rel, amax, tol)
    assert max_abs <= tol[0] * ref and (tol[1] is None or max_rel <= tol[1]), \
        f"{name}: max abs {max_abs:.3e} (scale {ref:.2f}) max rel {max_rel:.3e} tol {tol}"
    return max_abs, max_rel
