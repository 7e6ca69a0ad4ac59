"""GPU parity, kernel level: the reference's known-answer tests replayed through the C-ABI op entry
points (which launch the same HIP kernels the model path uses), plus randomized comparisons against
the CPU oracle.  Tolerances are the reference's own per-kernel table (runtime/ops/tolerance.go:13-24)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_kat.json")) as f:
    KAT = {c["name"]: c for c in json.load(f)["cases"]}

TOL = {"linear": 1e-4, "layer_norm": 1e-4, "rope": 2e-4, "attention": 2e-4, "conv1d": 2e-4, "convtranspose1d": 2e-4}


def seq(n):
    i = np.arange(n)
    return (((i % 17) - 8).astype(np.float32) / np.float32(17)).astype(np.float32)


def arr(c, key, shape_key=None):
    v = c[key]
    a = seq(int(v.split(":")[1])) if isinstance(v, str) else np.array(v, np.float32)
    return a.reshape(c[shape_key]) if shape_key and shape_key in c else a


def close(got, want, tol, rel=0.0):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert got.shape == want.shape, (got.shape, want.shape)
    err = np.abs(got - want)
    lim = tol + rel * np.abs(want)
    assert np.all(err <= lim), f"max err {err.max():.3e} (tol {tol}, rel {rel})"


@pytest.fixture(scope="module")
def R(pkg):
    return pkg.runtime


def test_linear_kat(R):
    c = KAT["linear_bias"]
    close(R.op_linear(arr(c, "x", "x_shape"), arr(c, "w", "w_shape"), c["bias"]).ravel(), c["want"], 1e-6)


@pytest.mark.parametrize("rows,inp,out", [(1, 1024, 3072), (3, 32, 1024), (64, 1024, 1024), (70, 100, 37), (5, 4096, 1024), (200, 192, 1)])
def test_linear_random_vs_oracle(R, rows, inp, out):
    rng = np.random.default_rng(rows * 7 + out)
    x = rng.standard_normal((rows, inp)).astype(np.float32)
    w = (rng.standard_normal((out, inp)) / np.sqrt(inp)).astype(np.float32)
    b = rng.standard_normal(out).astype(np.float32)
    close(R.op_linear(x, w, b), O.linear(x, w, b), TOL["linear"], TOL["linear"])


def test_layernorm_kat(R):
    c = KAT["layernorm_1234"]
    close(R.op_layernorm(np.array(c["x"], np.float32).reshape(c["shape"]), c["w"], c["b"], c["eps"]).ravel(), c["want"], c["tol"])


@pytest.mark.parametrize("rows,d,eps", [(1, 1024, 1e-5), (7, 512, 1e-6), (130, 100, 1e-5)])
def test_layernorm_random_vs_oracle(R, rows, d, eps):
    rng = np.random.default_rng(d)
    x = (rng.standard_normal((rows, d)) * 3 + 0.5).astype(np.float32)
    w, b = rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)
    close(R.op_layernorm(x, w, b, eps), O.layernorm(x, w, b, eps), TOL["layer_norm"], TOL["layer_norm"])
    close(R.op_layernorm(x, None, None, eps), O.layernorm(x, None, None, eps), TOL["layer_norm"], TOL["layer_norm"])


def test_rope_kat_and_errors(R, pkg):
    c = KAT["rope_quarter_turn"]
    x, cos, sin = arr(c, "x", "x_shape"), arr(c, "cos", "trig_shape"), arr(c, "sin", "trig_shape")
    close(R.op_rope(x, cos, sin, 0).ravel(), c["want"], c["tol"])
    with pytest.raises(pkg.PttsError, match="position must be >= 0"):
        R.op_rope(x, cos, sin, -1)
    with pytest.raises(pkg.PttsError, match="sequence length too small"):
        R.op_rope(x, cos, sin, 1)
    with pytest.raises(pkg.PttsError, match="must be even"):
        R.op_rope(np.zeros((1, 2, 3), np.float32), np.zeros((2, 1), np.float32), np.zeros((2, 1), np.float32), 0)


def test_rope_random_vs_oracle(R):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 16, 5, 64)).astype(np.float32)
    ang = rng.uniform(0, 6.28, (40, 32))
    cos, sin = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    close(R.op_rope(x, cos, sin, 17), O.rope(x, cos, sin, 17), TOL["rope"])


def test_attention_positions_kats(R):
    c = KAT["attention_positions_context_invalid_keys"]
    got = R.op_attention_positions(arr(c, "q", "q_shape"), arr(c, "k", "k_shape"), arr(c, "v", "v_shape"), c["posq"], c["posk"], c["context"])
    close(got.ravel(), c["want"], c["tol"])
    c = KAT["attention_positions_matches_causal_offset"]
    q, k = arr(c, "q", "q_shape"), arr(c, "k", "k_shape")
    v = np.zeros(k.shape, np.float32)   # the GPU op has dv == d: embed the reference's dv=3 values
    v3 = arr(c, "v", "v_shape")
    v[..., :3] = v3
    got = R.op_attention_positions(q, k, v, c["posq"], c["posk"], c["context"])
    close(got[..., :3], O.attention(q, k, v3, True, c["causal_offset"]), c["tol"])


def test_attention_nan_padding_never_read(R):
    # flow_transformer.go:404-420 + attention.go:402-406: slots beyond the valid length hold NaN in voice states
    rng = np.random.default_rng(5)
    q = rng.standard_normal((1, 2, 1, 64)).astype(np.float32)
    k = rng.standard_normal((1, 2, 6, 64)).astype(np.float32)
    v = rng.standard_normal((1, 2, 6, 64)).astype(np.float32)
    k[:, :, 4:] = np.nan
    v[:, :, 4:] = np.nan
    got = R.op_attention_positions(q, k, v, [3], [0, 1, 2, 3, -1, -1], -1)
    assert np.isfinite(got).all()
    close(got, O.attention_positions(q, k, v, [3], [0, 1, 2, 3, -1, -1], -1), TOL["attention"], TOL["attention"])


@pytest.mark.parametrize("tq,tk,ctx", [(1, 300, -1), (40, 40, -1), (300, 300, 250), (64, 64, 7)])
def test_attention_random_vs_oracle(R, tq, tk, ctx):
    rng = np.random.default_rng(tq + tk)
    q = rng.standard_normal((2, 3, tq, 64)).astype(np.float32)
    k = rng.standard_normal((2, 3, tk, 64)).astype(np.float32)
    v = rng.standard_normal((2, 3, tk, 64)).astype(np.float32)
    posq = list(range(tk - tq, tk))
    posk = list(range(tk))
    close(R.op_attention_positions(q, k, v, posq, posk, ctx), O.attention_positions(q, k, v, posq, posk, ctx), TOL["attention"], TOL["attention"])


def test_conv1d_leftpad_kats(R):
    c = KAT["conv1d_leftpad_matches_prepend"]
    x, w = arr(c, "x", "x_shape"), arr(c, "w", "w_shape")
    # the GPU op is the streaming form (stride 1, left pad k-1: mimi.go:69-76); compare with the oracle at those settings
    close(R.op_conv1d_leftpad(x, w, c["bias"]), O.conv1d(x, w, c["bias"], 1, w.shape[2] - 1, 0, 1, 1), 1e-5)
    c = KAT["conv1d_parallel_case"]
    x, w, b = arr(c, "x", "x_shape"), arr(c, "w", "w_shape"), arr(c, "bias")
    close(R.op_conv1d_leftpad(x, w, b), O.conv1d(x, w, b, 1, 2, 0, 1, 1), TOL["conv1d"])


@pytest.mark.parametrize("cin,cout,k,ln", [(512, 64, 7, 48), (64, 32, 3, 200), (32, 64, 1, 200), (16, 8, 3, 33)])
def test_conv1d_random_vs_oracle(R, cin, cout, k, ln):
    rng = np.random.default_rng(cin + k)
    x = rng.standard_normal((2, cin, ln)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    close(R.op_conv1d_leftpad(x, w, b), O.conv1d(x, w, b, 1, k - 1, 0, 1, 1), TOL["conv1d"], TOL["conv1d"])


def test_convtr_kats(R):
    c = KAT["convtr1d_ones"]   # k=2, stride 1: the streaming form keeps the first L*stride outputs (mimi.go:116-125)
    got = R.op_convtr1d_righttrim(arr(c, "x", "x_shape"), arr(c, "w", "w_shape"), None, 1, 1)
    close(got.ravel(), c["want"][:3], 0)
    c = KAT["convtr1d_depthwise"]
    got = R.op_convtr1d_righttrim(arr(c, "x", "x_shape"), arr(c, "w", "w_shape"), c["bias"], 1, 2)
    want = np.array(c["want"], np.float32).reshape(2, 4)[:, :3]
    close(got[0], want, 0)
    c = KAT["convtr1d_right_trim_matches_narrow"]   # [3,4,4] kernel, stride 2, trim 2 == k - stride
    x, w, b = arr(c, "x", "x_shape"), arr(c, "w", "w_shape"), arr(c, "bias")
    close(R.op_convtr1d_righttrim(x, w, b, 2, 1), O.convtr1d(x, w, b, 2, 0, 0, 1, 1, 2), 1e-5)


@pytest.mark.parametrize("cin,cout,stride,ln", [(64, 32, 6, 16), (32, 16, 5, 50), (16, 8, 4, 77)])
def test_convtr_random_vs_oracle(R, cin, cout, stride, ln):
    rng = np.random.default_rng(cin + stride)
    k = 2 * stride
    x = rng.standard_normal((2, cin, ln)).astype(np.float32)
    w = (rng.standard_normal((cin, cout, k)) / np.sqrt(2 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    close(R.op_convtr1d_righttrim(x, w, b, stride, 1), O.convtr1d(x, w, b, stride, 0, 0, 1, 1, k - stride), TOL["convtranspose1d"], TOL["convtranspose1d"])


def test_convtr_depthwise_random_vs_oracle(R):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 512, 5)).astype(np.float32)
    w = rng.standard_normal((512, 1, 32)).astype(np.float32)
    close(R.op_convtr1d_righttrim(x, w, None, 16, 512), O.convtr1d(x, w, None, 16, 0, 0, 1, 512, 16), TOL["convtranspose1d"], TOL["convtranspose1d"])


@pytest.mark.parametrize("n", [1, 7, 8, 9, 1000, 4099])
def test_pcm16_kernel_is_bit_exact(pkg, n):
    """wav_stream.go:43-54 on the device against the oracle: clamp, exact float64 product, truncation, NaN/inf, ragged tails."""
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) * 0.7).astype(np.float32)
    special = np.array([0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -3.0, np.nan, np.inf, -np.inf, 0.99999, -0.99999, 3.0517578e-05, -3.0517578e-05], np.float32)
    x[: min(n, special.size)] = special[: min(n, special.size)]
    assert np.array_equal(pkg.runtime.op_pcm16(x), O.pcm16(x))


@pytest.mark.parametrize("shape", [(4096, 256, 256, 0), (6000, 256, 256, 3), (16384, 640, 512, 0), (16384, 2048, 512, 1), (4100, 192, 224, 0)])
def test_weights_resident_gemm_equals_the_tile_gemm(pkg, shape):
    """k_gemm_wres (weights resident in LDS, barrier-free 32-row panels; the last two transposed convs and the Mimi linear1 at
    many rows) walks k in the same order as k_gemm3 and must give the same bits -- whole 256 x 256 tiles, 128-column tiles of a
    K = 512 product (N = 640: a ragged last tile), bias, ELU / GELU epilogues, a row count that is no multiple of 32, and a
    narrower N x K that is zero-padded in LDS."""
    import ctypes as C
    M, N, K, epi = shape
    L = pkg.runtime.hooks()
    L.ptts_debug_gemm.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 2
    us, md = C.c_float(0), C.c_float(-1)
    rc = L.ptts_debug_gemm(M, N, K, 1, 40, epi, 1, C.byref(us), C.byref(md))
    assert rc == 0, L.ptts_last_error().decode()
    assert md.value == 0.0, (shape, md.value)


@pytest.mark.parametrize("shape", [(16384, 1536, 512, 0x100), (16640, 1536, 512, 0x000), (16384, 512, 512, 0x204), (16384, 512, 2048, 0x204), (16384, 512, 3584, 0x003),
                                   (16384, 1536, 1024, 0x000), (49152, 128, 768, 0x403), (49152, 256, 128, 0x008), (16384, 2048, 512, 0x001), (16400, 256, 64, 0x000),
                                   (1600, 3072, 1024, 0x100), (1600, 1024, 1024, 0x204), (1600, 4096, 1024, 0x001), (1030, 128, 192, 0x003),
                                   (1600, 1024, 4096, 0x4000), (16384, 512, 2048, 0x4000)])
def test_many_row_gemm5_equals_gemm3_bit_for_bit_and_itself_run_to_run(pkg, shape):
    """k_gemm5 (bf16 weights; 256-row tiles from 16384 rows: the decoder's deep GEMMs at the benchmark's batch; 128 x 128 tiles from 1024 rows:
    the prompt prefill's projections, the last four shapes) keeps k_gemm3's k order, so every form the
    decoder uses must give k_gemm3's bits: the qkv projection with the RoPE epilogue (positions restarting every 2000 rows), plain bias,
    a residual read from the output buffer itself (out_proj / linear2 update the stream in place), ELU (first convolution), a prologue
    ELU (the residual block's first convolution: 128 columns), residual + ELU, GELU, a row count that is no multiple of the tile and a
    single weight chunk (K = 64), split-K in 1024-deep slices (the prefill's linear2: raw sums per plane).  The debug entry also runs the variant three more times and compares bits (a race would show)."""
    import ctypes as C
    M, N, K, epi = shape
    L = pkg.runtime.hooks()
    L.ptts_debug_gemm.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 2
    us, md = C.c_float(0), C.c_float(-1)
    rc = L.ptts_debug_gemm(M, N, K, 1, 50, epi, 1, C.byref(us), C.byref(md))
    assert rc == 0, L.ptts_last_error().decode()
    assert md.value == 0.0, (shape, md.value)


def test_many_row_gemm5_layer_scale_epilogue_rounds_like_the_reference(pkg):
    """residual + scale * (sums): k_gemm5 rounds the product and the sum one by one (the reference's r + s*v on amd64, mimi.go:275-285),
    k_gemm3 lets the compiler fuse them: the two may differ in the last bit, not more."""
    import ctypes as C
    L = pkg.runtime.hooks()
    L.ptts_debug_gemm.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 2
    us, md = C.c_float(0), C.c_float(-1)
    rc = L.ptts_debug_gemm(16384, 512, 512, 1, 50, 0xa05, 1, C.byref(us), C.byref(md))
    assert rc == 0, L.ptts_last_error().decode()
    assert 0.0 <= md.value <= 5e-7, md.value
