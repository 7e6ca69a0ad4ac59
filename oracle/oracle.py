"""ctypes binding + numpy file readers for the CPU ORACLE (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (go-pocket-tts_amd/) never does.

The numpy readers restate internal/safetensors/store.go (header parse :246-271, dtype
decode :339-395, float16 :397-431) and reader.go (voice-file classification :232-271,
model-state loading :273-308, embedding shape normalisation :219-230).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libptts_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compiles oracle/ptts_oracle.c with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "ptts_oracle.c")
    hdr = os.path.join(_HERE, "ptts_oracle.h")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


# --------------------------------------------------------------------------- safetensors (numpy)

def _f16_to_f32(bits: np.ndarray) -> np.ndarray:
    # store.go:397-431 handles zero/subnormal/inf/nan/normal explicitly == IEEE half -> single
    return bits.astype("<u2").view(np.float16).astype(np.float32)


def decode_tensor(raw: bytes, dtype: str, shape: list[int]) -> np.ndarray:
    """store.go:339-395: every dtype is decoded to float32 (I64 through float32(int64))."""
    n = int(np.prod(shape)) if len(shape) else 1
    if any(d == 0 for d in shape):
        n = 0
    dt = dtype.upper()
    if dt == "F32":
        if len(raw) < n * 4:
            raise ValueError(f"need {n*4} bytes for F32, got {len(raw)}")
        out = np.frombuffer(raw, "<f4", n).astype(np.float32)
    elif dt == "F16":
        if len(raw) < n * 2:
            raise ValueError(f"need {n*2} bytes for F16, got {len(raw)}")
        out = _f16_to_f32(np.frombuffer(raw, "<u2", n))
    elif dt == "BF16":
        if len(raw) < n * 2:
            raise ValueError(f"need {n*2} bytes for BF16, got {len(raw)}")
        out = (np.frombuffer(raw, "<u2", n).astype(np.uint32) << 16).view(np.float32)
    elif dt == "I64":
        if len(raw) < n * 8:
            raise ValueError(f"need {n*8} bytes for I64, got {len(raw)}")
        out = np.frombuffer(raw, "<i8", n).astype(np.float32)
    else:
        raise ValueError(f'unsupported dtype "{dtype}"')
    return out.reshape(shape)


class Store:
    """safetensors.Store (store.go:35-39,65-184)."""

    def __init__(self, data: bytes):
        if len(data) < 8:
            raise ValueError(f"safetensors: file too short ({len(data)} bytes)")
        (hlen,) = struct.unpack("<Q", data[:8])
        if 8 + hlen > len(data):
            raise ValueError(f"safetensors: header length {hlen} exceeds file size {len(data)}")
        try:
            header = json.loads(data[8:8 + hlen])
        except Exception as e:  # noqa: BLE001
            raise ValueError(f"safetensors: parse header: {e}") from e
        self.raw = data
        self.entries: dict[str, tuple[str, list[int], int, int]] = {}
        end_h = 8 + hlen
        for name in sorted(header):
            if name == "__metadata__":
                continue
            e = header[name]
            dt = str(e["dtype"]).upper()
            if dt not in ("F32", "F16", "BF16", "I64"):
                raise ValueError(f'safetensors: tensor "{name}" has unsupported dtype "{e["dtype"]}"')
            o0, o1 = e["data_offsets"]
            if o0 < 0 or o1 < o0:
                raise ValueError(f'safetensors: tensor "{name}" has invalid data offsets')
            shape = [int(d) for d in e["shape"]]
            if any(d < 0 for d in shape):
                raise ValueError(f'safetensors: tensor "{name}" has negative shape dimension')
            s, t = end_h + o0, end_h + o1
            if t > len(data):
                raise ValueError(f'safetensors: tensor "{name}" data [{s}:{t}] exceeds file size {len(data)}')
            n = 0 if any(d == 0 for d in shape) else int(np.prod(shape)) if shape else 1
            eb = {"F32": 4, "F16": 2, "BF16": 2, "I64": 8}[dt]
            if t - s < n * eb:
                raise ValueError(f'safetensors: tensor "{name}" needs {n*eb} bytes but data has {t-s}')
            self.entries[name.strip()] = (dt, shape, s, t)
        if not self.entries:
            raise ValueError("safetensors: no tensors found")
        self.names = sorted(self.entries)

    @staticmethod
    def open(path: str) -> "Store":
        with open(path, "rb") as f:
            return Store(f.read())

    def tensor(self, name: str) -> np.ndarray:
        if name not in self.entries:
            raise KeyError(f'safetensors: tensor "{name}" not found')
        dt, shape, s, t = self.entries[name]
        return decode_tensor(self.raw[s:t], dt, shape)

    def read_all(self) -> dict[str, np.ndarray]:
        return {n: self.tensor(n) for n in self.names}


def is_model_state_name(name: str) -> bool:  # reader.go:258-271
    slash = name.rfind("/")
    if slash <= 0 or slash == len(name) - 1:
        return False
    return name[slash + 1:] in ("cache", "offset", "current_end")


def classify_voice(names: list[str]) -> str:  # reader.go:232-256
    has_prompt = any(n == "audio_prompt" for n in names)
    has_state = any(is_model_state_name(n) for n in names if n != "audio_prompt")
    if has_state:
        return "model_state"
    if has_prompt or names:
        return "embedding"
    return "unknown"


def load_voice_embedding(store: Store) -> np.ndarray:  # reader.go:69-85,219-230
    if classify_voice(store.names) == "model_state":
        raise ValueError("safetensors: voice file contains upstream model state, not a legacy audio_prompt embedding")
    t = store.tensor(store.names[0])
    if t.ndim == 2:
        return t.reshape(1, *t.shape)
    if t.ndim == 3:
        return t
    raise ValueError(f"safetensors: voice embedding has {t.ndim}D shape {list(t.shape)}, expected 2D or 3D")


def load_voice_model_state(store: Store) -> dict[str, dict[str, np.ndarray]]:  # reader.go:273-308
    if classify_voice(store.names) != "model_state":
        raise ValueError("safetensors: voice file kind is not upstream model state")
    mods: dict[str, dict[str, np.ndarray]] = {}
    for name in store.names:
        slash = name.rfind("/")
        if slash <= 0 or slash == len(name) - 1:
            raise ValueError(f'safetensors: invalid model-state tensor name "{name}"')
        mod, key = name[:slash], name[slash + 1:]
        t = store.tensor(name)
        if key == "current_end":
            key, t = "offset", np.array([float(t.shape[0] if t.ndim else 0)], np.float32)
        mods.setdefault(mod, {})[key] = t
    return mods


# --------------------------------------------------------------------------- ctypes

class _PoTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)),
                ("shape", C.POINTER(C.c_int64)), ("rank", C.c_int32)]


class _PoRequest(C.Structure):
    _fields_ = [("tokens", C.POINTER(C.c_int64)), ("n_tokens", C.c_int64),
                ("temperature", C.c_float), ("eos_threshold", C.c_float),
                ("max_steps", C.c_int32), ("lsd_steps", C.c_int32), ("frames_after_eos", C.c_int32),
                ("voice_emb", C.POINTER(C.c_float)), ("voice_t", C.c_int64),
                ("voice_caches", C.POINTER(C.POINTER(C.c_float))),
                ("voice_steps", C.POINTER(C.c_int64)), ("voice_offsets", C.POINTER(C.c_int64)),
                ("noise", C.POINTER(C.c_float))]


class _PoResult(C.Structure):
    _fields_ = [("pcm", C.POINTER(C.c_float)), ("n_samples", C.c_int64),
                ("latents", C.POINTER(C.c_float)), ("n_frames", C.c_int32), ("eos_step", C.c_int32),
                ("eos_logits", C.POINTER(C.c_float))]


_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int64)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.po_dot.restype = C.c_float
        L.po_dot_generic.restype = C.c_float
        L.po_dot_avx2_order.restype = C.c_float
        L.po_dot_avx2_emul.restype = C.c_float
        for f in ("po_dot", "po_dot_generic", "po_dot_avx2_order", "po_dot_avx2_emul"):
            getattr(L, f).argtypes = [_FP, _FP, C.c_int64]
        L.po_model_create.restype = C.c_void_p
        L.po_model_create.argtypes = [C.POINTER(_PoTensor), C.c_int32, C.c_char_p, C.c_int32]
        L.po_model_free.argtypes = [C.c_void_p]
        L.po_state_new.restype = C.c_void_p
        L.po_state_new.argtypes = [C.c_void_p]
        L.po_state_from_voice.restype = C.c_void_p
        L.po_state_from_voice.argtypes = [C.c_void_p, C.POINTER(_FP), _IP, _IP, C.c_char_p, C.c_int32]
        L.po_state_free.argtypes = [C.c_void_p]
        L.po_state_offset.restype = C.c_int64
        L.po_state_offset.argtypes = [C.c_void_p, C.c_int]
        L.po_state_read_kv.argtypes = [C.c_void_p, C.c_int, _FP, _FP]
        L.po_mimi_out_len.restype = C.c_int64
        L.po_mimi_out_len.argtypes = [C.c_void_p, C.c_int64]
        L.po_conv1d_outlen.restype = C.c_int64
        L.po_convtr1d_outlen.restype = C.c_int64
        L.po_conv1d_outlen.argtypes = [C.c_int64] * 6
        L.po_convtr1d_outlen.argtypes = [C.c_int64] * 7
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_FP)


def _ip(a: np.ndarray):
    return a.ctypes.data_as(_IP)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def set_workers(tensor_workers: int, conv_workers: int) -> None:
    lib().po_set_workers(C.c_int(tensor_workers), C.c_int(conv_workers))


def set_use_avx2(on: bool) -> None:
    lib().po_set_use_avx2(C.c_int(1 if on else 0))


# ---- op-level wrappers (known-answer tests) ----

def dot(a, b, mode: str = "auto") -> float:
    a, b = _f32(a), _f32(b)
    fn = {"auto": lib().po_dot, "generic": lib().po_dot_generic, "avx2": lib().po_dot_avx2_order,
          "avx2_emul": lib().po_dot_avx2_emul}[mode]
    return float(fn(_fp(a), _fp(b), C.c_int64(a.size)))


def axpy(dst, alpha: float, src) -> np.ndarray:
    d, s = _f32(dst).copy(), _f32(src)
    lib().po_axpy(_fp(d), C.c_int64(d.size), C.c_float(alpha), _fp(s), C.c_int64(s.size))
    return d


def softmax(x) -> np.ndarray:
    x = _f32(x)
    y = np.empty_like(x)
    d = x.shape[-1]
    rc = lib().po_softmax_lastdim(_fp(x), C.c_int64(x.size // d), C.c_int64(d), _fp(y))
    if rc:
        raise ValueError("tensor: softmax encountered zero normalization sum")
    return y


def layernorm(x, w, b, eps: float) -> np.ndarray:
    x = _f32(x)
    y = np.empty_like(x)
    d = x.shape[-1]
    wa = _f32(w) if w is not None else None
    ba = _f32(b) if b is not None else None
    rc = lib().po_layernorm(_fp(x), _fp(wa) if wa is not None else None, _fp(ba) if ba is not None else None,
                            C.c_float(eps), C.c_int64(x.size // d), C.c_int64(d), _fp(y))
    if rc:
        raise ValueError("tensor: layernorm invalid")
    return y


def linear(x, w, bias=None) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    out, inp = w.shape
    batch = x.size // inp
    y = np.empty(x.shape[:-1] + (out,), np.float32)
    ba = _f32(bias) if bias is not None else None
    lib().po_linear(_fp(x), _fp(w), _fp(ba) if ba is not None else None,
                    C.c_int64(batch), C.c_int64(inp), C.c_int64(out), _fp(y))
    return y


def matmul2d(a, b) -> np.ndarray:
    a, b = _f32(a), _f32(b)
    m, k = a.shape
    n = b.shape[1]
    c = np.empty((m, n), np.float32)
    lib().po_matmul2d(_fp(a), _fp(b), C.c_int64(m), C.c_int64(k), C.c_int64(n), _fp(c))
    return c


def rope(x, cos, sin, pos: int) -> np.ndarray:
    x = _f32(x).copy()
    cos, sin = _f32(cos), _f32(sin)
    seq, dim = x.shape[-2], x.shape[-1]
    if pos < 0:
        raise ValueError("ops: rope position must be >= 0")
    if dim % 2:
        raise ValueError("ops: rope last dimension must be even")
    if cos.shape[0] < pos + seq:
        raise ValueError("ops: rope cos/sin sequence length too small")
    lib().po_rope(_fp(x), _fp(cos), _fp(sin), C.c_int64(x.size // (seq * dim)), C.c_int64(seq), C.c_int64(dim), C.c_int64(pos))
    return x


def attention(q, k, v, causal: bool, offset: int) -> np.ndarray:
    q, k, v = _f32(q), _f32(k), _f32(v)
    b, h, tq, d = q.shape
    tk, dv = k.shape[2], v.shape[3]
    out = np.empty((b, h, tq, dv), np.float32)
    rc = lib().po_attention(_fp(q), _fp(k), _fp(v), *[C.c_int64(i) for i in (b, h, tq, tk, d, dv)],
                            C.c_int(1 if causal else 0), C.c_int64(offset), _fp(out))
    if rc:
        raise ValueError("ops: attention failed")
    return out


def attention_positions(q, k, v, posq, posk, context: int) -> np.ndarray:
    q, k, v = _f32(q), _f32(k), _f32(v)
    b, h, tq, d = q.shape
    tk, dv = k.shape[2], v.shape[3]
    pq = np.ascontiguousarray(posq, np.int64)
    pk = np.ascontiguousarray(posk, np.int64)
    if pq.size != tq or pk.size != tk:
        raise ValueError("ops: attention pos length mismatch")
    out = np.empty((b, h, tq, dv), np.float32)
    rc = lib().po_attention_positions(_fp(q), _fp(k), _fp(v), *[C.c_int64(i) for i in (b, h, tq, tk, d, dv)],
                                      _ip(pq), _ip(pk), C.c_int64(context), _fp(out))
    if rc:
        raise ValueError("ops: softmax encountered zero normalization sum")
    return out


def conv1d(x, w, bias, stride=1, lpad=0, rpad=0, dilation=1, groups=1) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    b, ic, ln = x.shape
    oc, _, k = w.shape
    ol = int(lib().po_conv1d_outlen(ln, k, stride, lpad, rpad, dilation))
    if ol <= 0:
        raise ValueError("ops: conv1d produced non-positive output length")
    out = np.zeros((b, oc, ol), np.float32)
    ba = _f32(bias) if bias is not None else None
    rc = lib().po_conv1d(_fp(x), _fp(w), _fp(ba) if ba is not None else None,
                         *[C.c_int64(i) for i in (b, ic, ln, oc, k, stride, lpad, rpad, dilation, groups)], _fp(out))
    if rc:
        raise ValueError("ops: conv1d failed")
    return out


def convtr1d(x, w, bias, stride=1, pad=0, outpad=0, dilation=1, groups=1, right_trim=0) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    b, ic, ln = x.shape
    _, opg, k = w.shape
    ol = int(lib().po_convtr1d_outlen(ln, k, stride, pad, outpad, dilation, right_trim))
    if ol <= 0:
        raise ValueError("ops: convtranspose1d produced non-positive output length")
    out = np.zeros((b, opg * groups, ol), np.float32)
    ba = _f32(bias) if bias is not None else None
    rc = lib().po_convtr1d(_fp(x), _fp(w), _fp(ba) if ba is not None else None,
                           *[C.c_int64(i) for i in (b, ic, ln, opg, k, stride, pad, outpad, dilation, groups, right_trim)],
                           _fp(out))
    if rc:
        raise ValueError("ops: convtranspose1d failed")
    return out


def repack_convtr_kernel(w) -> np.ndarray:
    w = _f32(w)
    ic, oc, k = w.shape
    out = np.empty((k, oc, ic), np.float32)
    lib().po_repack_convtr_kernel(_fp(w), C.c_int64(ic), C.c_int64(oc), C.c_int64(k), _fp(out))
    return out


def mlp_silu(x, w1, b1, w2, b2) -> np.ndarray:
    x, w1, w2 = _f32(x), _f32(w1), _f32(w2)
    hid, inp = w1.shape
    out = w2.shape[0]
    batch = x.size // inp
    y = np.empty((batch, out), np.float32)
    b1a = _f32(b1) if b1 is not None else None
    b2a = _f32(b2) if b2 is not None else None
    lib().po_mlp_silu(_fp(x), _fp(w1), _fp(b1a) if b1a is not None else None, _fp(w2),
                      _fp(b2a) if b2a is not None else None,
                      C.c_int64(batch), C.c_int64(inp), C.c_int64(hid), C.c_int64(out), _fp(y))
    return y


def _inplace(fn, x):
    x = _f32(x).copy()
    fn(_fp(x), C.c_int64(x.size))
    return x


def gelu_erf(x): return _inplace(lib().po_gelu_erf, x)
def silu(x): return _inplace(lib().po_silu, x)
def elu(x): return _inplace(lib().po_elu, x)


def rmsnorm_alpha(x, alpha, eps: float) -> np.ndarray:
    x = _f32(x).copy()
    a = _f32(alpha)
    d = x.shape[-1]
    lib().po_rmsnorm_alpha(_fp(x), _fp(a), C.c_float(eps), C.c_int64(x.size // d), C.c_int64(d))
    return x


def replace_nan(x, vec) -> np.ndarray:
    x = _f32(x).copy()
    v = _f32(vec)
    lib().po_replace_nan(_fp(x), C.c_int64(x.size), _fp(v), C.c_int64(v.size))
    return x


def denorm_latent_to_bct(latent, std, mean) -> np.ndarray:
    latent = _f32(latent)
    b, t, d = latent.shape
    out = np.empty((b, d, t), np.float32)
    lib().po_denorm_latent_to_bct(_fp(latent), _fp(_f32(std)), _fp(_f32(mean)), C.c_int64(b), C.c_int64(t), C.c_int64(d), _fp(out))
    return out


def split_voice_kv(cache) -> tuple[np.ndarray, np.ndarray]:
    cache = _f32(cache)
    _, b, steps, heads, hd = cache.shape
    k = np.empty((b, heads, steps, hd), np.float32)
    v = np.empty_like(k)
    lib().po_split_voice_kv(_fp(cache), C.c_int64(b), C.c_int64(steps), C.c_int64(heads), C.c_int64(hd), _fp(k), _fp(v))
    return k, v


def read_voice_offset(t: np.ndarray) -> int:
    """flow_transformer.go:554-566."""
    if t.size == 0:
        raise ValueError("native: voice model state has empty offset tensor")
    v = np.float32(t.reshape(-1)[0])
    i = int(v)
    if np.float32(i) != v:
        raise ValueError(f"native: voice model state offset {v} is not an integer")
    return i


def voice_state_layers(modules, n_layers: int, heads: int = 0, head_dim: int = 0):
    """flowTransformer.initStateFromVoiceModelState up to the re-layout (flow_transformer.go:451-480): per layer
    layerStateFromVoiceModule (:517-552) with readVoiceStateOffset (:554-566) and splitVoiceKVCache's shape checks (:568-590),
    in the reference's order.  Returns (caches [2,B,T,H,D], steps, offsets); heads / head_dim 0 = unchecked (a layer without them)."""
    if modules is None:
        raise ValueError("native: voice model state is nil")
    caches, steps, offs = [], [], []
    for i in range(n_layers):
        name = f"transformer.layers.{i}.self_attn"  # flow_transformer.go:513-515
        mod = modules.get(name)
        if mod is None:
            raise ValueError(f'native: voice model state missing module "{name}"')
        if "cache" not in mod:
            raise ValueError(f'native: voice model state module "{name}" missing cache')
        if "offset" not in mod:
            raise ValueError(f'native: voice model state module "{name}" missing offset')
        try:
            off = read_voice_offset(np.asarray(mod["offset"], np.float32))
        except ValueError as e:
            raise ValueError(f'native: voice model state module "{name}" ' + str(e).replace("native: voice model state ", "")) from None
        c = _f32(mod["cache"])
        if c.ndim != 5:
            raise ValueError(f'native: voice model state module "{name}" cache shape {list(c.shape)}, want [2,B,T,H,D]')
        if c.shape[0] != 2:
            raise ValueError(f'native: voice model state module "{name}" cache first dim {c.shape[0]}, want 2')
        _, b, t, h, d = c.shape
        if b <= 0 or h <= 0 or d <= 0:
            raise ValueError(f'native: voice model state module "{name}" has invalid cache shape {list(c.shape)}')
        if heads and h != heads:
            raise ValueError(f'native: voice model state module "{name}" heads {h}, want {heads}')
        if head_dim and d != head_dim:
            raise ValueError(f'native: voice model state module "{name}" head dim {d}, want {head_dim}')
        if off < 0:
            raise ValueError(f'native: voice model state module "{name}" has negative offset {off}')
        if off > t:
            raise ValueError(f'native: voice model state module "{name}" offset {off} exceeds cache length {t}')
        caches.append(c)
        steps.append(t)
        offs.append(off)
    return caches, steps, offs


# ---- model-level wrapper ----

class OracleModel:
    """native.Model (model.go:25-138) on the CPU oracle."""

    def __init__(self, tensors: dict[str, np.ndarray]):
        L = lib()
        self._keep = []
        arr = (_PoTensor * len(tensors))()
        for i, (name, a) in enumerate(sorted(tensors.items())):
            a = _f32(a)
            shp = np.array(a.shape, np.int64)
            self._keep += [a, shp]
            arr[i] = _PoTensor(name.encode(), _fp(a), _ip(shp), a.ndim)
        err = C.create_string_buffer(512)
        self.h = L.po_model_create(arr, len(tensors), err, 512)
        if not self.h:
            raise ValueError(err.value.decode())
        self._keep = []  # the model copied everything
        dims = np.zeros(8, np.int64)
        L.po_model_dims(C.c_void_p(self.h), _ip(dims))
        (self.d_model, self.heads, self.n_layers, self.ldim, self.flow_dim,
         self.flow_depth, self.mimi_dim, self.n_bins) = [int(x) for x in dims]
        self.head_dim = self.d_model // self.heads

    @staticmethod
    def from_file(path: str) -> "OracleModel":
        return OracleModel(Store.open(path).read_all())

    def close(self):
        if self.h:
            lib().po_model_free(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # -- state
    def new_state(self):
        return OracleState(self, lib().po_state_new(C.c_void_p(self.h)))

    def state_from_voice(self, modules: dict[str, dict[str, np.ndarray]]):
        caches, steps, offs = voice_state_layers(modules, self.n_layers, self.heads, self.head_dim)
        ptrs = (_FP * self.n_layers)(*[_fp(c) for c in caches])
        st = np.array(steps, np.int64)
        of = np.array(offs, np.int64)
        err = C.create_string_buffer(512)
        h = lib().po_state_from_voice(C.c_void_p(self.h), ptrs, _ip(st), _ip(of), err, 512)
        if not h:
            raise ValueError(err.value.decode())
        return OracleState(self, h)

    # -- forward pieces
    def text_embeddings(self, ids) -> np.ndarray:
        ids = np.ascontiguousarray(ids, np.int64)
        out = np.empty((ids.size, self.d_model), np.float32)
        err = C.create_string_buffer(512)
        if lib().po_text_embeddings(C.c_void_p(self.h), _ip(ids), C.c_int64(ids.size), _fp(out), err, 512):
            raise ValueError(err.value.decode())
        return out

    def prompt(self, state: "OracleState", emb) -> None:
        emb = _f32(emb).reshape(-1, self.d_model)
        if lib().po_prompt(C.c_void_p(self.h), C.c_void_p(state.h), _fp(emb), C.c_int64(emb.shape[0])):
            raise ValueError("native: prompt failed")

    def step(self, state: "OracleState", frame_in, lsd_steps=1, eos_threshold=-4.0, noise=None):
        fi = _f32(frame_in).reshape(self.ldim)
        fo = np.empty(self.ldim, np.float32)
        last = np.empty(self.d_model, np.float32)
        is_eos = C.c_int(0)
        logit = C.c_float(0)
        nz = _f32(noise).reshape(self.ldim) if noise is not None else None
        rc = lib().po_step(C.c_void_p(self.h), C.c_void_p(state.h), _fp(fi), C.c_int(lsd_steps), C.c_float(eos_threshold),
                           _fp(nz) if nz is not None else None, _fp(fo), C.byref(is_eos), C.byref(logit), _fp(last))
        if rc:
            raise ValueError("native: step failed")
        return fo, bool(is_eos.value), float(logit.value), last

    def flow_main(self, seq, text):
        seq = _f32(seq).reshape(-1, self.ldim)
        text = _f32(text).reshape(-1, self.d_model)
        last = np.empty(self.d_model, np.float32)
        logit = C.c_float(0)
        rc = lib().po_flow_main(C.c_void_p(self.h), _fp(seq), C.c_int64(seq.shape[0]), _fp(text), C.c_int64(text.shape[0]),
                                _fp(last), C.byref(logit))
        if rc:
            raise ValueError("native: flow_main failed")
        return last, float(logit.value)

    def flow_direction(self, c, s: float, t: float, x) -> np.ndarray:
        c, x = _f32(c).reshape(self.d_model), _f32(x).reshape(self.ldim)
        out = np.empty(self.ldim, np.float32)
        lib().po_flow_direction(C.c_void_p(self.h), _fp(c), C.c_float(s), C.c_float(t), _fp(x), _fp(out))
        return out

    def latent_to_mimi(self, latent) -> np.ndarray:
        latent = _f32(latent).reshape(-1, self.ldim)
        t = latent.shape[0]
        out = np.empty((self.mimi_dim, t), np.float32)
        lib().po_latent_to_mimi(C.c_void_p(self.h), _fp(latent), C.c_int64(t), _fp(out))
        return out

    def mimi_decode(self, x) -> np.ndarray:
        x = _f32(x).reshape(self.mimi_dim, -1)
        t = x.shape[1]
        n = int(lib().po_mimi_out_len(C.c_void_p(self.h), C.c_int64(t)))
        pcm = np.empty(n, np.float32)
        if lib().po_mimi_decode(C.c_void_p(self.h), _fp(x), C.c_int64(t), _fp(pcm)):
            raise ValueError("native: mimi decode failed")
        return pcm

    def mimi_transformer(self, x) -> np.ndarray:
        """Staged: upsample + decoder transformer layers (mimi.go:733-748), rows [16 T, mimi_dim]."""
        x = _f32(x).reshape(self.mimi_dim, -1)
        t = x.shape[1]
        out = np.empty((16 * t, self.mimi_dim), np.float32)
        if lib().po_mimi_transformer(C.c_void_p(self.h), _fp(x), C.c_int64(t), _fp(out)):
            raise ValueError("native: mimi transformer failed")
        return out

    def debug_set_mimi_context(self, context: int) -> None:
        lib().po_debug_set_mimi_context(C.c_void_p(self.h), C.c_int64(context))

    def generate(self, tokens, *, max_steps=0, eos_threshold=-4.0, lsd_steps=1, frames_after_eos=3,
                 voice_emb=None, voice_state=None, noise=None):
        """tts.Runtime.GenerateAudio (runtime_native_safetensors.go:52-238)."""
        tokens = np.ascontiguousarray(tokens, np.int64)
        rq = _PoRequest()
        rq.tokens, rq.n_tokens = _ip(tokens), tokens.size
        rq.temperature, rq.eos_threshold = 0.0, eos_threshold
        rq.max_steps, rq.lsd_steps, rq.frames_after_eos = max_steps, lsd_steps, frames_after_eos
        keep = [tokens]
        if voice_emb is not None:
            ve = _f32(voice_emb).reshape(-1, self.d_model)
            keep.append(ve)
            rq.voice_emb, rq.voice_t = _fp(ve), ve.shape[0]
        if voice_state is not None:
            caches, steps, offs = [], [], []
            for i in range(self.n_layers):
                mod = voice_state[f"transformer.layers.{i}.self_attn"]
                c = _f32(mod["cache"])
                caches.append(c)
                steps.append(c.shape[2])
                offs.append(read_voice_offset(mod["offset"]))
            ptrs = (_FP * self.n_layers)(*[_fp(c) for c in caches])
            st, of = np.array(steps, np.int64), np.array(offs, np.int64)
            keep += [caches, ptrs, st, of]
            rq.voice_caches, rq.voice_steps, rq.voice_offsets = ptrs, _ip(st), _ip(of)
        if noise is not None:
            nz = _f32(noise).reshape(-1, self.ldim)
            keep.append(nz)
            rq.noise = _fp(nz)
        res = _PoResult()
        err = C.create_string_buffer(512)
        if lib().po_generate(C.c_void_p(self.h), C.byref(rq), C.byref(res), err, 512):
            raise ValueError(err.value.decode())
        pcm = np.ctypeslib.as_array(res.pcm, (res.n_samples,)).copy()
        lat = np.ctypeslib.as_array(res.latents, (res.n_frames, self.ldim)).copy()
        logits = np.ctypeslib.as_array(res.eos_logits, (res.n_frames,)).copy()
        out = {"pcm": pcm, "latents": lat, "n_frames": int(res.n_frames), "eos_step": int(res.eos_step), "eos_logits": logits}
        lib().po_free_result(C.byref(res))
        return out


class OracleState:
    def __init__(self, model: OracleModel, handle):
        self.model, self.h = model, handle

    def offset(self, layer: int = 0) -> int:
        return int(lib().po_state_offset(C.c_void_p(self.h), C.c_int(layer)))

    def kv(self, layer: int):
        n = self.offset(layer)
        k = np.empty((self.model.heads, n, self.model.head_dim), np.float32)
        v = np.empty_like(k)
        lib().po_state_read_kv(C.c_void_p(self.h), C.c_int(layer), _fp(k), _fp(v))
        return k, v

    def __del__(self):
        try:
            if self.h:
                lib().po_state_free(C.c_void_p(self.h))
                self.h = None
        except Exception:  # noqa: BLE001
            pass


def pcm16(samples) -> np.ndarray:
    """audio.WritePCM16Samples (wav_stream.go:43-54) without the byte packing: int16 per sample."""
    x = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
    out = np.zeros(x.size, dtype=np.int16)
    L = lib()
    L.po_pcm16.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.po_pcm16.restype = None
    L.po_pcm16(x.ctypes.data, x.size, out.ctypes.data)
    return out


def wav_header_streaming() -> bytes:
    """audio.WriteWAVHeaderStreaming (wav_stream.go:15-41)."""
    buf = (C.c_uint8 * 44)()
    L = lib()
    L.po_wav_header_streaming.argtypes = [C.c_void_p]
    L.po_wav_header_streaming.restype = None
    L.po_wav_header_streaming(buf)
    return bytes(buf)
