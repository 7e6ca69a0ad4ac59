"""The AR step's linears at 128 rows and more (csrc/tall.hip: k_rowprep + k_tall), stand-alone against f64 numpy on the same operands.

What they replace: Linear.Forward (internal/native/linear.go:117-182) behind LayerNorm (linear.go:295-309) in a transformer layer of the flow LM
(flow_transformer.go:326-389), GELU by erf (tensor_util.go:84-94).  The end-to-end checks -- teacher-forced against the oracle at 128 rows, slot symmetry and
64-row agreement at 128 / 256 -- are tests/test_gpu_wide_batch.py; here every shape of the launch plan and the edges of the tiling (rows that fill no whole
32- or 64-row tile, a last column block with one 16-column tile, split-K planes, the bf16 hi / lo plane epilogue) are checked value by value.

Tolerance: activations enter the matrix pipe as bf16 hi + lo (|x - hi - lo| <= 2^-17 |x|), weights are bf16 exactly, sums are f32 over K <= 4096:
|error| <= 3e-5 * (sum_k |a_k w_k| + |bias| + |R|) bounds both with a margin of ~4 (the same bound the 64-row step linear is tested with)."""
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _parity import record

pytestmark = pytest.mark.gpu
TOL = 3e-5


def bf16_round(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def layernorm64(x, w, b, eps):
    x = x.astype(np.float64)
    mu = x.mean(axis=1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=1, keepdims=True)      # biased, linear.go:295-309
    return (x - mu) / np.sqrt(var + eps) * w.astype(np.float64) + b.astype(np.float64)


def gelu64(v):
    return 0.5 * v * (1.0 + np.vectorize(math.erf)(v / math.sqrt(2.0)))


def check(name, got, want, bound):
    err = np.abs(got.astype(np.float64) - want)
    worst = float((err / bound).max())
    record(name, float(err.max()), 0.0, float(np.abs(want).max()), (TOL, 0))
    assert np.isfinite(got).all() and worst <= 1.0, (name, float(err.max()), worst)


@pytest.mark.parametrize("rows,n_out", [(128, 3072), (256, 3072), (129, 1024), (200, 1040), (97, 64)])
def test_rowprep_then_tall_is_layernorm_then_linear(pkg, rows, n_out):
    """[sum of 3 split-K planes + bias + residual -> LayerNorm -> in_proj]: the updated rows bit for bit (f32 adds in the kernel's order), the product to the
    operand-split bound; 129 / 200 / 97 rows leave a ragged last row tile, 1040 columns a last column block of one 16-column tile."""
    rng = np.random.default_rng(rows * 7 + n_out)
    K = 1024
    x = rng.standard_normal((rows, K), dtype=np.float32)
    planes = (0.3 * rng.standard_normal((3, rows, K))).astype(np.float32)
    pbias = (0.1 * rng.standard_normal(K)).astype(np.float32)
    lw = (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    lb = (0.1 * rng.standard_normal(K)).astype(np.float32)
    w = bf16_round(0.05 * rng.standard_normal((n_out, K)))
    bias = rng.standard_normal(n_out).astype(np.float32)
    got, x_out = pkg.runtime.debug_tall_linear(x, w, bias=bias, ln=(lw, lb, 1e-5), planes=planes, pbias=pbias)
    xs = x + (((planes[0] + planes[1]) + planes[2]) + pbias)          # the kernel's order, f32
    assert np.array_equal(x_out, xs)
    y = layernorm64(xs, lw, lb, 1e-5)
    want = y @ w.astype(np.float64).T + bias
    bound = TOL * (np.abs(y) @ np.abs(w.astype(np.float64)).T + np.abs(bias)) + 1e-6
    check(f"k_rowprep + k_tall [{rows} x {n_out} x 1024], LayerNorm + 3 planes", got, want, bound)


@pytest.mark.parametrize("rows", [128, 160, 256])
def test_tall_gelu_through_the_bf16_planes(pkg, rows):
    """linear1 + GELU leaving as the hi / lo planes linear2 reads: hi + lo is the value to 2^-16."""
    rng = np.random.default_rng(rows)
    K, N = 1024, 4096
    x = rng.standard_normal((rows, K), dtype=np.float32)
    lw, lb = np.ones(K, np.float32), np.zeros(K, np.float32)
    w = bf16_round(0.04 * rng.standard_normal((N, K)))
    bias = (0.2 * rng.standard_normal(N)).astype(np.float32)
    got, _ = pkg.runtime.debug_tall_linear(x, w, bias=bias, ln=(lw, lb, 1e-5), epi=1, out_planes=True)
    y = layernorm64(x, lw, lb, 1e-5)
    pre = y @ w.astype(np.float64).T + bias
    want = gelu64(pre)
    bound = 1.2 * TOL * (np.abs(y) @ np.abs(w.astype(np.float64)).T + np.abs(bias)) + 2.0 ** -16 * np.abs(want) + 1e-6   # |gelu'| <= 1.13
    check(f"k_tall [{rows} x 4096 x 1024] + GELU -> bf16 hi/lo planes", got, want, bound)


@pytest.mark.parametrize("rows,splitk", [(128, 4), (256, 4), (131, 4), (192, 1), (128, 2)])
def test_tall_linear2_split_k_planes_and_residual(pkg, rows, splitk):
    """linear2 over K = 4096: the split-K planes (plane 0 = residual + bias + its slice's sums, the others raw sums) add up to R + x W^T + b; splitk 1 is the
    residual-add epilogue."""
    rng = np.random.default_rng(rows + splitk)
    K, N = 4096, 1024
    x = rng.standard_normal((rows, K), dtype=np.float32)
    w = bf16_round(0.03 * rng.standard_normal((N, K)))
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((rows, N), dtype=np.float32)
    got, _ = pkg.runtime.debug_tall_linear(x, w, bias=bias, residual=res, epi=0 if splitk > 1 else 4, splitk=splitk)
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    if splitk > 1:
        ks = K // splitk
        for z in range(splitk):
            part = x64[:, z * ks:(z + 1) * ks] @ w64[:, z * ks:(z + 1) * ks].T
            ab = np.abs(x64[:, z * ks:(z + 1) * ks]) @ np.abs(w64[:, z * ks:(z + 1) * ks]).T
            if z == 0:
                part = res + (part + bias)
                ab = ab + np.abs(bias) + np.abs(res)
            check(f"k_tall linear2 [{rows} x 1024 x 4096] split-K {splitk}, plane {z}", got[z], part, TOL * ab + 1e-6)
        got = got.astype(np.float64).sum(axis=0)
    want = res + (x64 @ w64.T + bias)
    bound = TOL * (np.abs(x64) @ np.abs(w64).T + np.abs(bias) + np.abs(res)) * (splitk if splitk > 1 else 1) + 1e-6
    check(f"k_tall linear2 [{rows} x 1024 x 4096] split-K {splitk}, sum", got, want, bound)


def test_tall_rejects_what_it_does_not_take(pkg):
    x = np.zeros((128, 1000), np.float32)        # K % 128 != 0
    with pytest.raises(pkg.PttsError):
        pkg.runtime.debug_tall_linear(x, np.zeros((64, 1000), np.float32))
    x = np.zeros((300, 1024), np.float32)        # more rows than a step takes
    with pytest.raises(pkg.PttsError):
        pkg.runtime.debug_tall_linear(x, np.zeros((64, 1024), np.float32))
    x = np.zeros((128, 768), np.float32)         # a LayerNorm width k_rowprep is not built for
    with pytest.raises(pkg.PttsError):
        pkg.runtime.debug_tall_linear(x, np.zeros((64, 768), np.float32), ln=(np.ones(768, np.float32), np.zeros(768, np.float32), 1e-5))
