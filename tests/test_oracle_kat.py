"""Pins the CPU oracle against the reference's own known-answer tests (tests/golden/reference_kat.json).

Each fixture case cites the Go test it was transcribed from.  CPU only (no GPU marker).
"""
import json
import os
import struct

import numpy as np
import pytest

from oracle import oracle as O
from oracle import text_prepare as TP

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_kat.json")) as f:
    KAT = {c["name"]: c for c in json.load(f)["cases"]}


def seq(n: int) -> np.ndarray:
    i = np.arange(n)
    return (((i % 17) - 8).astype(np.float32) / np.float32(17)).astype(np.float32)


def arr(case, key, shape_key=None):
    v = case[key]
    a = seq(int(v.split(":")[1])) if isinstance(v, str) else np.array(v, np.float32)
    if shape_key and shape_key in case:
        a = a.reshape(case[shape_key])
    return a


def close(got, want, tol):
    got, want = np.asarray(got, np.float32).ravel(), np.asarray(want, np.float32).ravel()
    assert got.shape == want.shape
    assert np.all(np.abs(got - want) <= tol), (got, want)


@pytest.mark.parametrize("name", [n for n, c in KAT.items() if c["op"] == "dot"])
def test_dot(name):
    c = KAT[name]
    for mode in ("auto", "generic", "avx2"):
        assert abs(O.dot(c["a"], c["b"], mode) - c["want"]) <= c["tol"]


def test_dot_orders_agree_and_avx2_emulation_matches_intrinsics_shape():
    rng = np.random.default_rng(0)
    for n in (8, 9, 31, 32, 33, 40, 64, 192, 1000, 1024, 4096):
        a, b = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
        ref = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
        assert abs(O.dot(a, b, "avx2") - ref) < 1e-3 * max(1.0, abs(ref))
        assert abs(O.dot(a, b, "generic") - ref) < 1e-3 * max(1.0, abs(ref))
        assert O.dot(a, b, "avx2") == O.dot(a, b, "avx2_emul")   # intrinsics and lane-by-lane restatement: same bits
    # n < 8 takes the generic path even with AVX2 on (dot_amd64.go:13-19)
    a, b = rng.standard_normal(7).astype(np.float32), rng.standard_normal(7).astype(np.float32)
    assert O.dot(a, b, "auto") == O.dot(a, b, "generic")


@pytest.mark.parametrize("name", [n for n, c in KAT.items() if c["op"] == "axpy"])
def test_axpy(name):
    c = KAT[name]
    close(O.axpy(c["dst"], c["alpha"], c["src_vec"]), c["want"], c["tol"])


def test_softmax():
    c = KAT["softmax_123"]
    close(O.softmax(c["x"]), c["want"], c["tol"])


def test_layernorm():
    c = KAT["layernorm_1234"]
    close(O.layernorm(np.array(c["x"], np.float32).reshape(c["shape"]), c["w"], c["b"], c["eps"]), c["want"], c["tol"])


def test_matmul():
    c = KAT["matmul_2d"]
    close(O.matmul2d(arr(c, "a", "a_shape"), arr(c, "b", "b_shape")), c["want"], c["tol"])


def test_linear():
    c = KAT["linear_bias"]
    close(O.linear(arr(c, "x", "x_shape"), arr(c, "w", "w_shape"), c["bias"]), c["want"], c["tol"])


def test_rope():
    c = KAT["rope_quarter_turn"]
    got = O.rope(arr(c, "x", "x_shape"), arr(c, "cos", "trig_shape"), arr(c, "sin", "trig_shape"), c["pos"])
    close(got, c["want"], c["tol"])
    with pytest.raises(ValueError):  # rope_test.go TestRoPEErrors: negative position
        O.rope(arr(c, "x", "x_shape"), arr(c, "cos", "trig_shape"), arr(c, "sin", "trig_shape"), -1)
    with pytest.raises(ValueError):  # table too short
        O.rope(arr(c, "x", "x_shape"), arr(c, "cos", "trig_shape"), arr(c, "sin", "trig_shape"), 1)


def test_mlp():
    c = KAT["mlp_silu"]
    close(O.mlp_silu(arr(c, "x", "x_shape"), arr(c, "w1", "w1_shape"), None, arr(c, "w2", "w2_shape"), None), c["want"], c["tol"])


def test_attention_causal():
    c = KAT["attention_causal_masks_future"]
    got = O.attention(arr(c, "q", "q_shape"), arr(c, "k", "k_shape"), arr(c, "v", "v_shape"), c["causal"], c["offset"]).ravel()
    assert abs(got[0] - c["want_first"]) <= c["tol"]
    assert got[1] > c["want_second_gt"]


def test_attention_positions_context_and_invalid_keys():
    c = KAT["attention_positions_context_invalid_keys"]
    got = O.attention_positions(arr(c, "q", "q_shape"), arr(c, "k", "k_shape"), arr(c, "v", "v_shape"),
                                c["posq"], c["posk"], c["context"])
    close(got, c["want"], c["tol"])


def test_attention_positions_matches_causal_offset():
    c = KAT["attention_positions_matches_causal_offset"]
    q, k, v = arr(c, "q", "q_shape"), arr(c, "k", "k_shape"), arr(c, "v", "v_shape")
    close(O.attention_positions(q, k, v, c["posq"], c["posk"], c["context"]),
          O.attention(q, k, v, True, c["causal_offset"]), c["tol"])


def _attention_generic(q, k, v, causal, offset):
    """attentionGeneric (attention.go:88-129): MatMul, scale+mask+softmax, MatMul -- numpy f32/f64."""
    d = q.shape[-1]
    s = np.einsum("bhqd,bhkd->bhqk", q, k).astype(np.float32) * np.float32(1.0 / np.sqrt(d))
    tq, tk = s.shape[-2:]
    if causal:
        mask = np.arange(tk)[None, :] > (np.arange(tq)[:, None] + offset)
        s = np.where(mask, -np.inf, s)
    m = s.max(-1, keepdims=True)
    e = np.exp((s - m).astype(np.float64))
    p = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    return np.einsum("bhqk,bhkd->bhqd", p, v).astype(np.float32)


@pytest.mark.parametrize("name", ["attention4d_matches_generic_causal", "attention4d_matches_generic_noncausal"])
def test_attention_matches_generic(name):
    c = KAT[name]
    q, k, v = arr(c, "q", "q_shape"), arr(c, "k", "k_shape"), arr(c, "v", "v_shape")
    close(O.attention(q, k, v, c["causal"], c["offset"]), _attention_generic(q, k, v, c["causal"], c["offset"]), c["tol"])


def test_attention_fully_masked_row_is_zero_not_nan():
    # attention.go:423-425
    q = np.ones((1, 1, 1, 2), np.float32)
    k = np.full((1, 1, 2, 2), np.nan, np.float32)   # NaN padding must never be touched (:402-406)
    v = np.full((1, 1, 2, 2), np.nan, np.float32)
    out = O.attention_positions(q, k, v, [0], [-1, -1], -1)
    assert np.all(out == 0)


@pytest.mark.parametrize("name", [n for n, c in KAT.items() if c["op"] == "conv1d"])
def test_conv1d(name):
    c = KAT[name]
    got = O.conv1d(arr(c, "x", "x_shape"), arr(c, "w", "w_shape"), c.get("bias"), c["stride"], c["lpad"], c["rpad"],
                   c["dilation"], c["groups"])
    close(got, c["want"], c["tol"])


def test_conv1d_leftpad_matches_prepend():
    c = KAT["conv1d_leftpad_matches_prepend"]
    x, w = arr(c, "x", "x_shape"), arr(c, "w", "w_shape")
    got = O.conv1d(x, w, c["bias"], c["stride"], c["lpad"], 0, c["dilation"], 1)
    padded = np.concatenate([np.zeros(x.shape[:2] + (c["lpad"],), np.float32), x], axis=2)
    close(got, O.conv1d(padded, w, c["bias"], c["stride"], 0, 0, c["dilation"], 1), c["tol"])


def test_conv1d_workers_invariant():
    c = KAT["conv1d_parallel_case"]
    x, w, b = arr(c, "x", "x_shape"), arr(c, "w", "w_shape"), arr(c, "bias")
    O.set_workers(1, 4)
    got = O.conv1d(x, w, b, c["stride"], c["lpad"], c["rpad"], c["dilation"], c["groups"])
    O.set_workers(1, 1)
    close(got, O.conv1d(x, w, b, c["stride"], c["lpad"], c["rpad"], c["dilation"], c["groups"]), c["tol"])


@pytest.mark.parametrize("name", [n for n, c in KAT.items() if c["op"] == "convtr1d"])
def test_convtr1d(name):
    c = KAT[name]
    got = O.convtr1d(arr(c, "x", "x_shape"), arr(c, "w", "w_shape"), c.get("bias"), c["stride"], 0, 0, 1,
                     c["groups"], c["right_trim"])
    close(got, c["want"], c["tol"])


def test_convtr_repack():
    c = KAT["convtr1d_repack"]
    close(O.repack_convtr_kernel(arr(c, "w", "w_shape")), c["want"], c["tol"])


@pytest.mark.parametrize("name", ["convtr1d_right_trim_matches_narrow", "convtr1d_prepacked_right_trim_matches_narrow"])
def test_convtr_right_trim(name):
    c = KAT[name]
    x, w, b = arr(c, "x", "x_shape"), arr(c, "w", "w_shape"), arr(c, "bias")
    got = O.convtr1d(x, w, b, c["stride"], 0, 0, 1, c["groups"], c["right_trim"])
    full = O.convtr1d(x, w, b, c["stride"], 0, 0, 1, c["groups"], 0)
    close(got, full[:, :, : full.shape[2] - c["right_trim"]], c["tol"])


def test_convtr_workers_invariant():
    c = KAT["convtr1d_parallel_case"]
    x, w, b = arr(c, "x", "x_shape"), arr(c, "w", "w_shape"), arr(c, "bias")
    O.set_workers(1, 4)
    got = O.convtr1d(x, w, b, c["stride"], 0, 0, 1, c["groups"], 0)
    O.set_workers(1, 1)
    close(got, O.convtr1d(x, w, b, c["stride"], 0, 0, 1, c["groups"], 0), c["tol"])


def test_denorm_latent_to_bct():
    c = KAT["denorm_latent_to_bct"]
    lat = arr(c, "latent", "latent_shape")
    std, mean = np.array(c["std"], np.float32), np.array(c["mean"], np.float32)
    want = (lat * std + mean).transpose(0, 2, 1)
    close(O.denorm_latent_to_bct(lat, std, mean), want, c["tol"])


def _self_attention(c, x, context, stateful_split=None):
    """mimi.go:365-441 / flow_transformer.go:194-324 on one tiny layer, composed from oracle ops."""
    H, hd = c["heads"], c["head_dim"]
    w_in, w_out = arr(c, "in_proj", "in_proj_shape"), arr(c, "out_proj", "out_proj_shape")
    cos, sin = arr(c, "cos", "trig_shape"), arr(c, "sin", "trig_shape")
    b, t, d = x.shape

    def project(xs, pos):
        qkv = O.linear(xs, w_in)
        tt = xs.shape[1]
        q, k, v = [qkv[..., i * d:(i + 1) * d].reshape(b, tt, H, hd).transpose(0, 2, 1, 3) for i in range(3)]
        return O.rope(q, cos, sin, pos), O.rope(k, cos, sin, pos), np.ascontiguousarray(v)

    if stateful_split is None:
        q, k, v = project(x, 0)
        pos = list(range(t))
        a = O.attention_positions(q, k, v, pos, pos, context) if context >= 0 else O.attention(q, k, v, True, 0)
    else:
        _, k0, v0 = project(x[:, :stateful_split], 0)
        q, k1, v1 = project(x[:, stateful_split:], stateful_split)
        k, v = np.concatenate([k0, k1], 2), np.concatenate([v0, v1], 2)
        a = O.attention_positions(q, k, v, [stateful_split], list(range(t)), -1)
    tt = a.shape[2]
    return O.linear(np.ascontiguousarray(a.transpose(0, 2, 1, 3)).reshape(b, tt, d), w_out)


def test_mimi_self_attention_uses_context_window():
    c = KAT["mimi_self_attention_context_window"]
    close(_self_attention(c, arr(c, "x", "x_shape"), c["context"]), c["want"], c["tol"])


def test_stateful_attention_matches_full_last_token():
    c = KAT["stateful_attention_matches_full_last_token"]
    x = arr(c, "x", "x_shape")
    full = _self_attention(c, x, -1)
    got = _self_attention(c, x, -1, stateful_split=2)
    close(got, full[:, 2:], c["tol"])


def test_latent_to_mimi_projector_matches_denorm_conv():
    c = KAT["latent_to_mimi_projector"]
    std, mean = np.array(c["std"], np.float32), np.array(c["mean"], np.float32)
    w, bias, lat = arr(c, "w", "w_shape"), np.array(c["bias"], np.float32), arr(c, "latent", "latent_shape")
    want = O.conv1d(O.denorm_latent_to_bct(lat, std, mean), w, bias, 1, 0, 0, 1, 1)
    # model.go:226-242 fold, then model.go:294-303 projection
    w2 = w[:, :, 0]
    wf = (w2 * std[None, :]).astype(np.float32)
    bf = bias.copy()
    for oc in range(w2.shape[0]):
        bv = bias[oc]
        for ic in range(w2.shape[1]):
            bv = np.float32(bv + np.float32(w2[oc, ic] * mean[ic]))
        bf[oc] = bv
    got = np.stack([[O.dot(lat[0, t], wf[oc]) + bf[oc] for t in range(lat.shape[1])] for oc in range(w2.shape[0])])
    close(got, want, c["tol"])


def test_split_voice_kv():
    c = KAT["voice_state_kv_relayout"]
    k, v = O.split_voice_kv(arr(c, "cache", "cache_shape"))
    close(k, c["want_k"], 0)
    close(v, c["want_v"], 0)


def test_voice_offset_must_be_integral():
    assert O.read_voice_offset(np.array([3.0], np.float32)) == 3
    with pytest.raises(ValueError):
        O.read_voice_offset(np.array([2.5], np.float32))


def test_dtype_decode():
    c = KAT["dtype_decode_f16_bf16"]
    f16 = struct.pack("<3H", *c["f16_bits"])
    close(O.decode_tensor(f16, "F16", [3]), c["want"], c["tol"])
    bf = (np.array(c["bf16_from_f32"], np.float32).view(np.uint32) >> 16).astype("<u2").tobytes()
    close(O.decode_tensor(bf, "BF16", [3]), c["want"], c["tol"])
    close(O.decode_tensor(struct.pack("<2q", 5, -7), "I64", [2]), [5, -7], 0)
    with pytest.raises(ValueError):
        O.decode_tensor(b"\0\0", "F32", [1])
    with pytest.raises(ValueError):
        O.decode_tensor(b"\0\0\0\0", "F64", [1])


def test_estimate_max_frames_and_frames_after_eos():
    for tokens, rate, want in KAT["estimate_max_frames"]["table"]:
        assert TP.estimate_max_frames(tokens, rate) == want
    for words, want in KAT["frames_after_eos"]["table"]:
        assert TP.frames_after_eos(words) == want


def test_prepare_chunks_upstream_cases():
    enc = lambda s: list(range(1, len(s.split()) + 1))  # stubTokenizer prepare_test.go:9-21
    for inp, want_first in KAT["prepare_text_first_chunk" if False else "prepare_chunks_upstream_cases"]["table"]:
        chunks = TP.prepare_chunks(inp, enc, 50)
        assert chunks[0]["text"] == want_first
        assert TP.frames_after_eos(chunks[0]["num_words"]) > 0
    with pytest.raises(ValueError):
        TP.prepare_chunks("   ", enc, 50)


@pytest.mark.parametrize("name", ["pcm16_encoding", "pcm16_clamping"])
def test_pcm16(name):
    """audio/wav_stream_test.go:106-149: clamp to [-1, 1], * 32767, truncation; the reference's own tolerance is 1 LSB."""
    c = KAT[name]
    got = O.pcm16(np.array(c["samples"], np.float32))
    assert np.abs(got.astype(np.int32) - np.array(c["want"], np.int32)).max() <= c["tol_lsb"]


def test_pcm16_truncates_toward_zero_and_maps_nan_to_zero():
    x = np.array([0.99999, -0.99999, 1e-5, -1e-5, np.nan, np.inf, -np.inf, 0.25], np.float32)
    want = [int(np.trunc(np.float64(v) * 32767.0)) if np.isfinite(v) else (0 if np.isnan(v) else int(np.sign(v)) * 32767) for v in x]
    assert O.pcm16(x).tolist() == want


def test_wav_header_streaming():
    c = KAT["wav_header_streaming"]
    h = O.wav_header_streaming()
    assert len(h) == c["len"]
    assert h[0:4] == c["riff"].encode() and h[8:12] == c["wave"].encode() and h[12:16] == c["fmt"].encode() and h[36:40] == c["data"].encode()
    u32 = lambda o: int.from_bytes(h[o:o + 4], "little")
    u16 = lambda o: int.from_bytes(h[o:o + 2], "little")
    assert u32(4) == c["riff_size"] and u32(40) == c["data_size"] and u32(16) == c["fmt_size"]
    assert (u16(20), u16(22), u32(24), u32(28), u16(32), u16(34)) == (c["format"], c["channels"], c["sample_rate"], c["byte_rate"], c["block_align"], c["bits"])
