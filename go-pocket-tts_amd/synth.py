"""Synthetic PocketTTS checkpoints, voices and prompts (numpy only).

There are no real weights in the build/bench environment, so every
parity test and benchmark runs on a generated safetensors file that carries the
exact tensor names and shapes the reference loaders ask for
(internal/native/flow_lm.go:51-119, flow_transformer.go:110-156,482-511,
flow_net.go:18-40,92-113,181-203,250-305, mimi.go:546-637, conditioner.go:17).

The file format is the one internal/safetensors/store.go:246-271 parses and
writer.go:15 emits: 8-byte LE header length, JSON header, raw little-endian data.
"""
from __future__ import annotations

import json
import struct
from dataclasses import dataclass, asdict

import numpy as np


@dataclass(frozen=True)
class SynthConfig:
    """Shapes of checkpoint b6369a24 by default (PLAN.md:35-39, mimi.go:25-33)."""
    d_model: int = 1024        # 16 heads x 64 (heads are fixed by DefaultFlowLMConfig, flow_lm.go:20-27)
    n_layers: int = 6
    ffn: int = 4096
    n_bins: int = 4000         # table has n_bins + 1 rows
    ldim: int = 32
    flow_dim: int = 512
    flow_depth: int = 6
    freq_embed: int = 256      # timestep frequency embedding (cos|sin) width
    mimi_dim: int = 512        # fixed by the depthwise upsample groups=512 (mimi.go:567)
    mimi_layers: int = 2
    mimi_ffn: int = 2048
    n_filters: int = 64        # SEANet ladder 8f -> 4f -> 2f -> f -> 1
    input_linear_bias: bool = False
    speaker_proj: bool = False  # also write flow_lm.speaker_proj_weight [d_model, 512]
    layer_scale: float = 0.01  # Mimi layer_scale_{1,2}; ~1 makes the decoder transformer's branches count in full

    @staticmethod
    def full() -> "SynthConfig":
        return SynthConfig()

    @staticmethod
    def tiny() -> "SynthConfig":
        """Same head geometry (head_dim 64), far fewer/lighter layers: CPU-oracle friendly."""
        return SynthConfig(n_layers=2, ffn=512, n_bins=63, flow_dim=128, flow_depth=2,
                           freq_embed=64, mimi_layers=1, mimi_ffn=256, n_filters=16,
                           input_linear_bias=True)


def _bf16_round(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even f32 -> bf16 bit pattern (uint16)."""
    u = a.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) >> 16).astype(np.uint16)


def bf16_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << 16).view(np.float32)


def _f16_bits(a: np.ndarray) -> np.ndarray:
    return a.astype(np.float16).view(np.uint16)


def write_safetensors(path: str, tensors: dict[str, np.ndarray], dtype: str = "F32",
                      per_tensor_dtype: dict[str, str] | None = None) -> None:
    """Writes tensors in sorted-name order.  dtype: F32 | BF16 | F16 for float arrays;
    int64 arrays are always stored as I64 (store.go:381-391)."""
    header: dict[str, dict] = {}
    blobs: list[bytes] = []
    off = 0
    for name in sorted(tensors):
        a = tensors[name]
        dt = (per_tensor_dtype or {}).get(name, dtype)
        if a.dtype == np.int64:
            raw, dt = a.astype("<i8").tobytes(), "I64"
        elif dt == "F32":
            raw = a.astype("<f4").tobytes()
        elif dt == "BF16":
            raw = _bf16_round(a).astype("<u2").tobytes()
        elif dt == "F16":
            raw = _f16_bits(a).astype("<u2").tobytes()
        else:
            raise ValueError(f"unsupported dtype {dt}")
        header[name] = {"dtype": dt, "shape": list(a.shape), "data_offsets": [off, off + len(raw)]}
        blobs.append(raw)
        off += len(raw)
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for b in blobs:
            f.write(b)


def make_checkpoint(cfg: SynthConfig = SynthConfig(), seed: int = 1234) -> dict[str, np.ndarray]:
    """Random-init tensors of the reference architecture: weights
    ~ N(0, 1/fan_in), norm weights 1 +- 0.1, emb_std in [0.5, 1.5], layer_scale 0.01."""
    rng = np.random.default_rng(seed)
    t: dict[str, np.ndarray] = {}

    def lin(name: str, out: int, inp: int, bias: bool, gain: float = 1.0) -> None:
        t[name + ".weight"] = (rng.standard_normal((out, inp)) * (gain / np.sqrt(inp))).astype(np.float32)
        if bias:
            t[name + ".bias"] = (rng.standard_normal(out) * 0.02).astype(np.float32)

    def norm(name: str, d: int) -> None:
        t[name + ".weight"] = (1.0 + 0.1 * rng.uniform(-1, 1, d)).astype(np.float32)
        t[name + ".bias"] = (0.02 * rng.standard_normal(d)).astype(np.float32)

    D, F = cfg.d_model, cfg.ffn
    t["flow_lm.conditioner.embed.weight"] = rng.standard_normal((cfg.n_bins + 1, D)).astype(np.float32)
    t["flow_lm.emb_std"] = rng.uniform(0.5, 1.5, cfg.ldim).astype(np.float32)
    t["flow_lm.emb_mean"] = (0.1 * rng.standard_normal(cfg.ldim)).astype(np.float32)
    t["flow_lm.bos_emb"] = rng.standard_normal(cfg.ldim).astype(np.float32)
    lin("flow_lm.input_linear", D, cfg.ldim, cfg.input_linear_bias)
    norm("flow_lm.out_norm", D)
    lin("flow_lm.out_eos", 1, D, True)
    for i in range(cfg.n_layers):
        p = f"flow_lm.transformer.layers.{i}"
        norm(p + ".norm1", D)
        norm(p + ".norm2", D)
        lin(p + ".self_attn.in_proj", 3 * D, D, False)
        lin(p + ".self_attn.out_proj", D, D, False, gain=0.5)
        lin(p + ".linear1", F, D, False)
        lin(p + ".linear2", D, F, False, gain=0.5)
    C = cfg.flow_dim
    half = cfg.freq_embed // 2
    for i in range(2):
        p = f"flow_lm.flow_net.time_embed.{i}"
        t[p + ".freqs"] = np.exp(-np.log(10000.0) * np.arange(half, dtype=np.float64) / half).astype(np.float32)
        lin(p + ".mlp.0", C, cfg.freq_embed, True)
        lin(p + ".mlp.2", C, C, True)
        t[p + ".mlp.3.alpha"] = (1.0 + 0.1 * rng.uniform(-1, 1, C)).astype(np.float32)
    lin("flow_lm.flow_net.cond_embed", C, D, True)
    lin("flow_lm.flow_net.input_proj", C, cfg.ldim, True)
    for i in range(cfg.flow_depth):
        p = f"flow_lm.flow_net.res_blocks.{i}"
        norm(p + ".in_ln", C)
        lin(p + ".mlp.0", C, C, True)
        lin(p + ".mlp.2", C, C, True)
        lin(p + ".adaLN_modulation.1", 3 * C, C, True, gain=0.5)
    lin("flow_lm.flow_net.final_layer.linear", cfg.ldim, C, True)
    if cfg.speaker_proj:   # voice cloning: Mimi-encoder latents [T, 512] -> [T, d_model] (onnx/voice_encode.go:174-188)
        t["flow_lm.speaker_proj_weight"] = (rng.standard_normal((D, 512)) / np.sqrt(512.0)).astype(np.float32)
    lin("flow_lm.flow_net.final_layer.adaLN_modulation.1", 2 * C, C, True, gain=0.5)

    M = cfg.mimi_dim
    t["mimi.quantizer.output_proj.weight"] = (rng.standard_normal((M, cfg.ldim, 1)) / np.sqrt(cfg.ldim)).astype(np.float32)
    t["mimi.upsample.convtr.convtr.weight"] = (rng.standard_normal((M, 1, 32)) * 0.5).astype(np.float32)
    for i in range(cfg.mimi_layers):
        p = f"mimi.decoder_transformer.transformer.layers.{i}"
        norm(p + ".norm1", M)
        norm(p + ".norm2", M)
        lin(p + ".self_attn.in_proj", 3 * M, M, False)
        lin(p + ".self_attn.out_proj", M, M, False)
        lin(p + ".linear1", cfg.mimi_ffn, M, False)
        lin(p + ".linear2", M, cfg.mimi_ffn, False)
        if cfg.layer_scale == 0.01:
            t[p + ".layer_scale_1.scale"] = np.full(M, 0.01, np.float32)
            t[p + ".layer_scale_2.scale"] = np.full(M, 0.01, np.float32)
        else:   # per-channel values around the requested scale (a uniform vector would hide a channel mix-up)
            t[p + ".layer_scale_1.scale"] = (cfg.layer_scale * (1.0 + 0.25 * rng.uniform(-1, 1, M))).astype(np.float32)
            t[p + ".layer_scale_2.scale"] = (cfg.layer_scale * (1.0 + 0.25 * rng.uniform(-1, 1, M))).astype(np.float32)

    def conv(name: str, oc: int, ic: int, k: int) -> None:
        t[name + ".weight"] = (rng.standard_normal((oc, ic, k)) / np.sqrt(ic * k)).astype(np.float32)
        t[name + ".bias"] = (0.02 * rng.standard_normal(oc)).astype(np.float32)

    def convtr(name: str, ic: int, oc: int, k: int) -> None:
        # each output sample sums 2 taps x ic channels
        t[name + ".weight"] = (rng.standard_normal((ic, oc, k)) / np.sqrt(2 * ic)).astype(np.float32)
        t[name + ".bias"] = (0.02 * rng.standard_normal(oc)).astype(np.float32)

    f = cfg.n_filters
    ch = [8 * f, 4 * f, 2 * f, f]
    conv("mimi.decoder.model.0.conv", ch[0], M, 7)
    for j, (idx_up, idx_rb, stride) in enumerate(((2, 3, 6), (5, 6, 5), (8, 9, 4))):
        convtr(f"mimi.decoder.model.{idx_up}.convtr", ch[j], ch[j + 1], 2 * stride)
        conv(f"mimi.decoder.model.{idx_rb}.block.1.conv", ch[j + 1] // 2, ch[j + 1], 3)
        conv(f"mimi.decoder.model.{idx_rb}.block.3.conv", ch[j + 1], ch[j + 1] // 2, 1)
    conv("mimi.decoder.model.11.conv", 1, ch[3], 3)
    return t


INT8_STEP_MATRICES = ("self_attn.in_proj.weight", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight")


def int8_step_weight_names(tensors: dict[str, np.ndarray]) -> list[str]:
    """The matrices PTTS_WEIGHTS_INT8 quantizes: everything the AR step streams (model.cpp step_linear / ada_all)."""
    names = []
    for k in tensors:
        if not k.endswith(".weight") or tensors[k].ndim != 2:
            continue
        if k.startswith("flow_lm.transformer.layers.") and k.endswith(INT8_STEP_MATRICES):
            names.append(k)
        elif k in ("flow_lm.input_linear.weight", "flow_lm.out_eos.weight"):
            names.append(k)
        elif k.startswith("flow_lm.flow_net.") and ".time_embed." not in k:
            names.append(k)
    return sorted(names)


def quantize_int8_rows(w: np.ndarray) -> np.ndarray:
    """W^ = q * s with s = max|row| / 127 (1 for a zero row), q = rint(W / s) clipped to [-127, 127] -- the same f32 arithmetic as
    model.cpp quantize_rows, so the result is bit-identical to the weights the library computes with."""
    w = w.astype(np.float32)
    mx = np.abs(w).max(axis=1)
    s = np.where(mx > 0, mx / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    q = np.clip(np.rint(w / s[:, None]), -127, 127).astype(np.float32)
    return (q * s[:, None]).astype(np.float32)


def dequantized_int8_checkpoint(tensors: dict[str, np.ndarray]) -> dict[str, np.ndarray]:
    """The checkpoint an f32 reference must be given to compute with the weights of PTTS_WEIGHTS_INT8."""
    out = dict(tensors)
    for k in int8_step_weight_names(tensors):
        out[k] = quantize_int8_rows(tensors[k])
    return out


def quantize_like_file(tensors: dict[str, np.ndarray], dtype: str) -> dict[str, np.ndarray]:
    """What a reader sees after the file round trip (BF16/F16 storage loses mantissa bits)."""
    if dtype == "F32":
        return {k: v.astype(np.float32) for k, v in tensors.items()}
    if dtype == "BF16":
        return {k: bf16_to_f32(_bf16_round(v)).reshape(v.shape) for k, v in tensors.items()}
    if dtype == "F16":
        return {k: v.astype(np.float16).astype(np.float32) for k, v in tensors.items()}
    raise ValueError(dtype)


def make_voice_state(cfg: SynthConfig, offset: int = 125, capacity: int | None = None,
                     seed: int = 7, legacy_current_end: bool = False) -> dict[str, np.ndarray]:
    """Synthetic upstream model-state voice: per layer `transformer.layers.{i}.self_attn/cache`
    F32 [2,1,T,16,64] with NaN beyond `offset`, and `.../offset` I64 [1]
    (internal/safetensors/reader.go:232-308, flow_transformer.go:513-552)."""
    rng = np.random.default_rng(seed)
    T = capacity if capacity is not None else offset
    heads, hd = 16, cfg.d_model // 16
    out: dict[str, np.ndarray] = {}
    for i in range(cfg.n_layers):
        c = (rng.standard_normal((2, 1, T, heads, hd)) * 0.7).astype(np.float32)
        c[:, :, offset:] = np.nan
        mod = f"transformer.layers.{i}.self_attn"
        out[mod + "/cache"] = c
        if legacy_current_end:
            out[mod + "/current_end"] = np.zeros(offset, np.float32)
        else:
            out[mod + "/offset"] = np.array([offset], np.int64)
    return out


def make_voice_embedding(cfg: SynthConfig, frames: int = 125, seed: int = 11) -> dict[str, np.ndarray]:
    """Legacy `audio_prompt` embedding [1, T, d_model] (reader.go:69-85)."""
    rng = np.random.default_rng(seed)
    return {"audio_prompt": rng.standard_normal((1, frames, cfg.d_model)).astype(np.float32)}


def make_prompts(n: int, tokens: int = 25, n_bins: int = 4000, seed: int = 42) -> np.ndarray:
    """Fixed-length synthetic prompts: ids uniform in [0, n_bins)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, n_bins, size=(n, tokens), dtype=np.int64)


def config_dict(cfg: SynthConfig) -> dict:
    return asdict(cfg)
