"""Host-side mirror of the reference's Runtime seam over the C ABI (include/ptts.h).

Reference: `tts.Runtime` (internal/tts/runtime.go:42-45), `RuntimeGenerateConfig` (:17-31),
`VoiceEmbedding` (:11-14), `nativeSafetensorsRuntime` (runtime_native_safetensors.go:20-244).
The reference toolchain (Go) is absent from the build image, so this ctypes layer plays the
role the cgo shim in INTEGRATION.md plays for the Go service: same method names, argument
meaning and error behaviour.  There is no CPU fallback: if libptts_hip.so is missing or no
HIP device is visible, every call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTTS_LIB_PATH") or os.path.join(_HERE, "libptts_hip.so")   # PTTS_LIB_PATH: A/B of two builds on one box (tools/gpu_r3.sh)

PTTS_OK, PTTS_EINVAL, PTTS_EIO, PTTS_EFORMAT, PTTS_ENODEVICE, PTTS_ECANCELLED, PTTS_ENOMEM = range(7)
WEIGHTS_F32, WEIGHTS_BF16, WEIGHTS_INT8 = 0, 1, 2
KV_F32, KV_BF16 = 0, 1

_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int64)
_STEP_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32)
_PCM_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p)


class PttsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


class Cancelled(PttsError):
    """ctx.Err() of the reference (runtime_native_safetensors.go:156-159)."""


class _Opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("weights", C.c_int32), ("kv", C.c_int32), ("max_batch", C.c_int32),
                ("use_graph", C.c_int32), ("reserved0", C.c_int32), ("reserved", C.c_int32 * 10)]


class _Info(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("d_model", "n_heads", "n_layers", "ffn", "ldim", "n_bins", "flow_dim", "flow_depth",
                                         "mimi_dim", "mimi_heads", "mimi_layers", "mimi_context", "sample_rate",
                                         "samples_per_frame", "steps_per_latent")] + \
               [("frame_rate", C.c_double), ("encoder_frame_rate", C.c_double), ("n_params", C.c_int64),
                ("arena_bytes", C.c_int64), ("weights", C.c_int32), ("kv", C.c_int32)]


class _Request(C.Structure):
    _fields_ = [("tokens", _IP), ("n_tokens", C.c_int64), ("temperature", C.c_float), ("eos_threshold", C.c_float),
                ("max_steps", C.c_int32), ("estimated_max_steps", C.c_int32), ("lsd_steps", C.c_int32),
                ("frames_after_eos", C.c_int32), ("voice_embedding", _FP), ("voice_frames", C.c_int64),
                ("voice_caches", C.POINTER(_FP)), ("voice_cache_steps", _IP), ("voice_offsets", _IP), ("noise", _FP),
                ("step_callback", _STEP_CB), ("callback_user", C.c_void_p), ("cancel", C.POINTER(C.c_int32)),
                ("want_latents", C.c_int32), ("pcm_format", C.c_int32), ("voice", C.c_void_p), ("noise_seed", C.c_uint64), ("noise_rows", C.c_int32), ("reserved", C.c_int32 * 1),
                ("pcm_callback", _PCM_CB), ("pcm_user", C.c_void_p), ("stream_frames", C.c_int32), ("reserved2", C.c_int32 * 3)]


class _Profile(C.Structure):
    _fields_ = [("launches", C.c_int64), ("total_ms", C.c_double), ("algorithmic_bytes", C.c_double), ("kernel", C.c_char * 64),
                ("weight_bytes", C.c_double), ("prefill_ms", C.c_double), ("ar_loop_ms", C.c_double), ("mimi_ms", C.c_double)]


class _Result(C.Structure):
    _fields_ = [("pcm", _FP), ("n_samples", C.c_int64), ("latents", _FP), ("n_frames", C.c_int32), ("eos_step", C.c_int32),
                ("status", C.c_int32), ("pcm16", C.POINTER(C.c_int16)), ("reserved", C.c_int32 * 2)]


# the same layout as a numpy record (generate_batch reads a whole batch of results as one table)
_RESULT_DTYPE = np.dtype({"names": ["pcm", "n_samples", "latents", "n_frames", "eos_step", "status", "pcm16", "reserved"],
                          "formats": ["<u8", "<i8", "<u8", "<i4", "<i4", "<i4", "<u8", ("<i4", 2)],
                          "offsets": [_Result.pcm.offset, _Result.n_samples.offset, _Result.latents.offset, _Result.n_frames.offset,
                                      _Result.eos_step.offset, _Result.status.offset, _Result.pcm16.offset, _Result.reserved.offset],
                          "itemsize": C.sizeof(_Result)})

class _VoiceTensor(C.Structure):   # ptts_voice_tensor
    _fields_ = [("data", C.POINTER(C.c_float)), ("count", C.c_int64), ("rank", C.c_int32), ("reserved", C.c_int32), ("shape", C.c_int64 * 8)]


_lib = None

# every symbol include/ptts.h declares (tests check that the built library exports all of them -- and none of HOOK_SYMBOLS)
ABI_SYMBOLS = [
    "ptts_default_opts", "ptts_model_open", "ptts_model_open_bytes", "ptts_model_close", "ptts_model_info", "ptts_last_error",
    "ptts_plan_create", "ptts_plan_create_bytes", "ptts_plan_arena_bytes", "ptts_model_open_planned", "ptts_plan_free",
    "ptts_generate", "ptts_free_result", "ptts_text_embeddings", "ptts_batch_new", "ptts_batch_free", "ptts_batch_reset",
    "ptts_batch_set_voice_state", "ptts_batch_prompt", "ptts_batch_step", "ptts_batch_offsets", "ptts_batch_read_kv",
    "ptts_decode_latents", "ptts_noise_rows", "ptts_speaker_project", "ptts_flow_direction", "ptts_op_linear", "ptts_op_layernorm", "ptts_op_rope",
    "ptts_op_attention_positions", "ptts_op_conv1d_leftpad", "ptts_op_convtr1d_righttrim", "ptts_version",
        "ptts_voice_create", "ptts_voice_free", "ptts_profile_enable", "ptts_profile_read", "ptts_plan_fill_host", "ptts_wav_header_streaming", "ptts_op_pcm16",
    "ptts_dispatcher_create", "ptts_dispatcher_create_custom", "ptts_dispatch_generate", "ptts_dispatcher_stats", "ptts_dispatcher_close",
    "ptts_model_share", "ptts_model_replicate", "ptts_model_set_use_graph", "ptts_model_set_max_batch", "ptts_text_estimate_max_frames", "ptts_text_frames_after_eos", "ptts_text_prepare", "ptts_text_chunks", "ptts_chunks_count",
    "ptts_chunks_get", "ptts_chunks_free",
    "ptts_tokenizer_open", "ptts_tokenizer_open_bytes", "ptts_tokenizer_free", "ptts_tokenizer_vocab_size", "ptts_tokenizer_encode",
    "ptts_tokenizer_encode_cb", "ptts_text_nfkc", "ptts_rccl_unique_id", "ptts_rccl_broadcast", "ptts_dsp_apply",
    "ptts_voice_file_open", "ptts_voice_file_open_bytes", "ptts_voice_file_close", "ptts_voice_file_kind", "ptts_voice_file_embedding",
    "ptts_voice_file_modules", "ptts_voice_file_module", "ptts_voice_file_state", "ptts_voice_open", "ptts_voice_open_bytes",
    ]
# the test / measurement hooks of include/ptts_debug.h: exported by libptts_hooks.so, never by libptts_hip.so (checked by __graft_entry__.build())
HOOK_SYMBOLS = [
    "ptts_decode_stages", "ptts_mimi_layer_piece", "ptts_debug_last_attention_kernel", "ptts_debug_launch_counts", "ptts_debug_flow_cluster_inject",
    "ptts_debug_time_skinny", "ptts_debug_skinny_stamps", "ptts_debug_gemm", "ptts_debug_step_stamps", "ptts_debug_tall_linear",
]


def build(force: bool = False) -> str:
    """Compiles the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", _HERE, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PttsError(PTTS_ENODEVICE, f"{LIB_PATH} is missing: run __graft_entry__.build() (no CPU fallback exists)")
        # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; two HSA runtimes in one process leave the
        # second without a GPU.  Importing torch first makes the dynamic linker resolve our DT_NEEDED libamdhip64.so.7
        # to the copy torch already loaded, so tests / bench.py (torch.distributed plumbing) and this library share one
        # runtime.  Hosts that never load torch (the Go service) set PTTS_NO_TORCH_PRELOAD=1 or simply lack torch.
        if "torch" not in sys.modules and not os.environ.get("PTTS_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except Exception:  # noqa: BLE001
                pass
        L = C.CDLL(LIB_PATH)
        L.ptts_last_error.restype = C.c_char_p
        L.ptts_version.restype = C.c_char_p
        L.ptts_plan_arena_bytes.restype = C.c_size_t
        L.ptts_plan_arena_bytes.argtypes = [C.c_void_p]
        L.ptts_plan_free.argtypes = [C.c_void_p]
        L.ptts_model_close.argtypes = [C.c_void_p]
        L.ptts_batch_free.argtypes = [C.c_void_p]
        L.ptts_model_open.argtypes = [C.c_char_p, C.POINTER(_Opts), C.POINTER(C.c_void_p)]
        L.ptts_model_open_bytes.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_Opts), C.POINTER(C.c_void_p)]
        L.ptts_plan_create.argtypes = [C.c_char_p, C.POINTER(_Opts), C.POINTER(C.c_void_p)]
        L.ptts_plan_create_bytes.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_Opts), C.POINTER(C.c_void_p)]
        L.ptts_model_open_planned.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.ptts_plan_fill_host.argtypes = [C.c_void_p, C.c_void_p]
        L.ptts_model_info.argtypes = [C.c_void_p, C.POINTER(_Info)]
        L.ptts_generate.argtypes = [C.c_void_p, C.POINTER(_Request), C.c_int32, C.POINTER(_Result)]
        L.ptts_free_result.argtypes = [C.POINTER(_Result)]
        L.ptts_text_embeddings.argtypes = [C.c_void_p, _IP, C.c_int64, _FP]
        L.ptts_batch_new.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.ptts_batch_reset.argtypes = [C.c_void_p]
        L.ptts_batch_set_voice_state.argtypes = [C.c_void_p, C.c_int32, C.POINTER(_FP), _IP, _IP]
        L.ptts_batch_prompt.argtypes = [C.c_void_p, _FP, _IP]
        L.ptts_batch_step.argtypes = [C.c_void_p, _FP, C.c_int32, _FP, _FP, _FP, _FP]
        L.ptts_batch_offsets.argtypes = [C.c_void_p, _IP]
        L.ptts_batch_read_kv.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _FP, _FP]
        L.ptts_decode_latents.argtypes = [C.c_void_p, _FP, C.c_int32, C.c_int32, _FP, _FP]
        L.ptts_noise_rows.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_int32, _FP]
        L.ptts_flow_direction.argtypes = [C.c_void_p, _FP, C.c_float, C.c_float, _FP, C.c_int32, _FP]
        L.ptts_voice_create.argtypes = [C.c_void_p, C.POINTER(_FP), _IP, _IP, C.POINTER(C.c_void_p)]
        L.ptts_voice_free.argtypes = [C.c_void_p]
        L.ptts_voice_file_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.ptts_voice_file_open_bytes.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.ptts_voice_file_close.argtypes = [C.c_void_p]
        L.ptts_voice_file_close.restype = None
        L.ptts_voice_file_kind.argtypes = [C.c_void_p]
        L.ptts_voice_file_kind.restype = C.c_int32
        L.ptts_voice_file_embedding.argtypes = [C.c_void_p, C.POINTER(_FP), _IP]
        L.ptts_voice_file_modules.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.ptts_voice_file_module.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(_VoiceTensor), C.POINTER(_VoiceTensor)]
        L.ptts_voice_file_state.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_FP), _IP, _IP]
        L.ptts_voice_open.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.ptts_voice_open_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.ptts_profile_enable.argtypes = [C.c_void_p, C.c_int32]
        L.ptts_profile_read.argtypes = [C.c_void_p, C.POINTER(_Profile)]
        L.ptts_op_linear.argtypes = [_FP, _FP, _FP, C.c_int64, C.c_int64, C.c_int64, _FP]
        L.ptts_op_layernorm.argtypes = [_FP, _FP, _FP, C.c_float, C.c_int64, C.c_int64, _FP]
        L.ptts_op_rope.argtypes = [_FP, _FP, _FP] + [C.c_int64] * 5
        L.ptts_op_attention_positions.argtypes = [_FP, _FP, _FP] + [C.c_int64] * 5 + [_IP, _IP, C.c_int64, _FP]
        L.ptts_op_conv1d_leftpad.argtypes = [_FP, _FP, _FP] + [C.c_int64] * 5 + [_FP]
        L.ptts_op_convtr1d_righttrim.argtypes = [_FP, _FP, _FP] + [C.c_int64] * 7 + [_FP]
        _lib = L
    return _lib


HOOKS_PATH = os.path.join(_HERE, "libptts_hooks.so")   # (also beside another build of the product library named by PTTS_LIB_PATH: it binds to the libptts_hip.so SONAME already loaded)
_hooks = None


def hooks():
    """libptts_hooks.so: the entry points of include/ptts_debug.h (launch census, staged decoder observation points, in-kernel stamps, micro-benchmarks,
    fault injection).  Loaded beside the product library by tests and tools only; it resolves its internals against the libptts_hip.so already mapped."""
    global _hooks
    if _hooks is None:
        lib()   # the product library first: the hooks' DT_NEEDED libptts_hip.so then resolves to the copy already in the process
        if not os.path.exists(HOOKS_PATH):
            raise PttsError(PTTS_ENODEVICE, f"{HOOKS_PATH} is missing: run __graft_entry__.build()")
        H = C.CDLL(HOOKS_PATH, mode=C.RTLD_GLOBAL)
        H.ptts_decode_stages.argtypes = [C.c_void_p, _FP, C.c_int32, C.c_int32, _FP, _FP, _FP]
        H.ptts_debug_last_attention_kernel.restype = C.c_char_p
        H.ptts_debug_launch_counts.restype = C.c_int64
        H.ptts_debug_launch_counts.argtypes = [C.c_int32, C.c_char_p, C.c_int64]
        H.ptts_mimi_layer_piece.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _FP, C.c_int64, C.c_int32, C.c_int32, _FP]
        H.ptts_debug_flow_cluster_inject.argtypes = [C.c_void_p, C.c_int32]
        _hooks = H
    return _hooks


def _check(rc: int):
    if rc != PTTS_OK:
        msg = lib().ptts_last_error().decode(errors="replace")
        raise (Cancelled if rc == PTTS_ECANCELLED else PttsError)(rc, msg)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _fp(a: Optional[np.ndarray]):
    return C.cast(a.ctypes.data, _FP) if a is not None else None   # (ndarray.ctypes.data_as is three times slower; 64 requests per call)


def _ip(a: np.ndarray):
    return C.cast(a.ctypes.data, _IP)


# ----------------------------------------------------------------------------- reference-shaped types

@dataclass
class VoiceEmbedding:
    """tts.VoiceEmbedding (runtime.go:11-14): shape [1, T, D]."""
    data: np.ndarray
    shape: Sequence[int]


@dataclass
class VoiceModelState:
    """safetensors.VoiceModelState (reader.go:29-31): modules[name]["cache" | "offset"]."""
    modules: dict


@dataclass
class RuntimeGenerateConfig:
    """tts.RuntimeGenerateConfig (runtime.go:17-31).  `noise` is this build's explicit form of the
    reference's rng draws (flow_lm.go:386-408): [max_steps, 32] rows of N(0,1)*sqrt(temperature)."""
    temperature: float = 0.0
    eos_threshold: float = -4.0
    max_steps: int = 0
    estimated_max_steps: int = 0
    lsd_decode_steps: int = 1
    frames_after_eos: int = 0
    mimi_steps_per_latent: int = 0   # accepted and ignored, like the native reference runtime
    mimi_sequence_length: int = 0
    voice_embedding: Optional[VoiceEmbedding] = None
    voice_model_state: Optional[VoiceModelState] = None
    device_voice: Optional["DeviceVoice"] = None   # a VoiceModelState already uploaded with Model.upload_voice
    step_callback: Optional[Callable[[int, int], None]] = None
    noise: Optional[np.ndarray] = None
    noise_seed: int = 0   # device draw (temperature > 0, no injected noise): 0 = a fresh stream per request, like the reference's clock-seeded rng
    cancel: Optional[np.ndarray] = None  # int32[1]; nonzero = cancelled (the ctx of GenerateAudio)
    want_latents: bool = False
    pcm16: bool = False   # PCM egress on the device: GenerateResult.pcm is int16 = audio.WritePCM16Samples (wav_stream.go:43-54)
    # frame-granular streaming: pcm_callback(sample_offset, samples) is called from a library thread for consecutive ranges of
    # `stream_frames` frames while generation is still running (samples: a copy, float32 or int16 per pcm16)
    pcm_callback: Optional[Callable[[int, np.ndarray], None]] = None
    stream_frames: int = 0


def _free_addr(addr: int):
    """Hand one result buffer back to the library's page-locked pool."""
    r = _Result()
    r.pcm = C.cast(C.c_void_p(addr), _FP)
    lib().ptts_free_result(C.byref(r))


class _OwnedBuffer:
    """A result buffer handed over by the library: exposes it to numpy without a copy and frees it on collection."""

    def __init__(self, ptr, n: int, typestr: str):
        self._addr = ptr if type(ptr) is int else C.cast(ptr, C.c_void_p).value
        self.__array_interface__ = {"shape": (n,), "typestr": typestr, "data": (self._addr, False), "version": 3}

    def __del__(self):
        if self._addr and sys is not None and not sys.is_finalizing():
            _free_addr(self._addr)
            self._addr = None


@dataclass
class GenerateResult:
    pcm: np.ndarray
    n_frames: int
    eos_step: int
    latents: Optional[np.ndarray] = None


@dataclass
class ModelInfo:
    d_model: int; n_heads: int; n_layers: int; ffn: int; ldim: int; n_bins: int
    flow_dim: int; flow_depth: int; mimi_dim: int; mimi_heads: int; mimi_layers: int; mimi_context: int
    sample_rate: int; samples_per_frame: int; steps_per_latent: int
    frame_rate: float; encoder_frame_rate: float; n_params: int; arena_bytes: int; weights: int; kv: int


def _opts(device=0, weights=WEIGHTS_F32, kv=KV_F32, max_batch=64, use_graph=False) -> _Opts:
    o = _Opts()
    lib().ptts_default_opts(C.byref(o))
    o.device, o.weights, o.kv, o.max_batch, o.use_graph = device, weights, kv, max_batch, 1 if use_graph else 0
    return o


class Model:
    """native.Model (internal/native/model.go:25-138) resident in HBM."""

    def __init__(self, handle: int):
        self.h = handle
        i = _Info()
        _check(lib().ptts_model_info(self.h, C.byref(i)))
        self.info = ModelInfo(**{n: getattr(i, n) for n, _ in _Info._fields_})

    @staticmethod
    def open(path: str, **kw) -> "Model":
        """LoadModelFromSafetensors (model.go:33-40)."""
        h = C.c_void_p()
        o = _opts(**kw)
        _check(lib().ptts_model_open(path.encode(), C.byref(o), C.byref(h)))
        return Model(h.value)

    @staticmethod
    def open_bytes(data: bytes, **kw) -> "Model":
        """LoadModelFromStore over OpenStoreFromBytes (store.go:65)."""
        h = C.c_void_p()
        o = _opts(**kw)
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        _check(lib().ptts_model_open_bytes(buf, len(data), C.byref(o), C.byref(h)))
        return Model(h.value)

    @staticmethod
    def plan(path: str, **kw) -> tuple[int, int]:
        """Returns (plan handle, arena bytes) for the two-phase multi-GPU open."""
        p = C.c_void_p()
        o = _opts(**kw)
        _check(lib().ptts_plan_create(path.encode(), C.byref(o), C.byref(p)))
        return p.value, int(lib().ptts_plan_arena_bytes(p.value))

    @staticmethod
    def plan_fill_host(plan: int, nbytes: int) -> np.ndarray:
        """Host image of the arena (decode + convert + derive), no GPU needed."""
        buf = np.zeros(nbytes, np.uint8)
        _check(lib().ptts_plan_fill_host(plan, buf.ctypes.data_as(C.c_void_p)))
        return buf

    @staticmethod
    def plan_free(plan: int):
        lib().ptts_plan_free(plan)

    @staticmethod
    def open_planned(plan: int, device_arena_ptr: int, fill: bool) -> "Model":
        h = C.c_void_p()
        _check(lib().ptts_model_open_planned(plan, C.c_void_p(device_arena_ptr), 1 if fill else 0, C.byref(h)))
        return Model(h.value)

    def set_use_graph(self, on: bool):
        """ptts_opts.use_graph of an open model: hipGraph replay of the AR step (True) or plain launches (False)."""
        lib().ptts_model_set_use_graph.argtypes = [C.c_void_p, C.c_int32]
        _check(lib().ptts_model_set_use_graph(self.h, 1 if on else 0))

    def set_max_batch(self, n: int):
        """ptts_opts.max_batch of an open model / engine (1..256): utterances of one call that are stepped together."""
        lib().ptts_model_set_max_batch.argtypes = [C.c_void_p, C.c_int32]
        _check(lib().ptts_model_set_max_batch(self.h, int(n)))

    def share(self) -> "Model":
        """A second engine over this model's weights (own streams, KV caches, workspaces); this model must outlive it."""
        h = C.c_void_p()
        lib().ptts_model_share.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        _check(lib().ptts_model_share(self.h, C.byref(h)))
        return Model(h.value)

    def replicate(self, device: int) -> "Model":
        """This model on another GPU of the process (own arena, filled from this one's by hipMemcpyPeer); independent of this one afterwards."""
        h = C.c_void_p()
        lib().ptts_model_replicate.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        _check(lib().ptts_model_replicate(self.h, int(device), C.byref(h)))
        return Model(h.value)

    def close(self):
        if self.h:
            lib().ptts_model_close(self.h)
            self.h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():   # the HIP runtime may already be gone at interpreter teardown
            return
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # -- staged methods
    def text_embeddings(self, ids) -> np.ndarray:
        ids = np.ascontiguousarray(ids, np.int64)
        out = np.empty((ids.size, self.info.d_model), np.float32)
        _check(lib().ptts_text_embeddings(self.h, _ip(ids), ids.size, _fp(out)))
        return out

    def upload_voice(self, state: VoiceModelState) -> "DeviceVoice":
        """Keeps a voice model state in HBM (the reference re-reads the file per Synthesize call: service.go:127,216-246)."""
        ptrs, steps, offs, arrs = _voice_arrays(state, self.info.n_layers)
        h = C.c_void_p()
        _check(lib().ptts_voice_create(self.h, ptrs, _ip(steps), _ip(offs), C.byref(h)))
        return DeviceVoice(h.value, int(offs[0]))

    def mimi_layer_qkv(self, layer: int, x, pos0: int = 0, rows_per_seg: int = 0) -> np.ndarray:
        """norm1 -> in_proj -> RoPE(q, k) of Mimi decoder-transformer layer `layer` on rows x [R, 512] -> [R, 1536] (ptts_mimi_layer_piece)."""
        x = _f32(x)
        out = np.empty((x.shape[0], 3 * x.shape[1]), np.float32)
        _check(hooks().ptts_mimi_layer_piece(self.h, layer, 0, _fp(x), x.shape[0], pos0, rows_per_seg, _fp(out)))
        return out

    def debug_flow_cluster_inject(self, block: int) -> None:
        """Test hook: the next plain-launched AR step's k_flow_cluster runs with one workgroup withholding its publish for flow-net block `block` (1-based)."""
        _check(hooks().ptts_debug_flow_cluster_inject(self.h, int(block)))

    def mimi_layer_ffn(self, layer: int, x) -> np.ndarray:
        """x + layer_scale_2 * linear2(gelu(linear1(norm2(x)))) of Mimi decoder-transformer layer `layer` on rows x [R, 512]."""
        x = _f32(x)
        out = np.empty_like(x)
        _check(hooks().ptts_mimi_layer_piece(self.h, layer, 1, _fp(x), x.shape[0], 0, 0, _fp(out)))
        return out

    def open_voice(self, src) -> "DeviceVoice":
        """A model-state voice file (path, or the file's bytes) straight into HBM: safetensors.LoadVoiceModelState +
        initStateFromVoiceModelState + the upload (ptts_voice_open / ptts_voice_open_bytes)."""
        h = C.c_void_p()
        if isinstance(src, (bytes, bytearray, memoryview)):
            buf = bytes(src)
            _check(lib().ptts_voice_open_bytes(self.h, buf, len(buf), C.byref(h)))
        else:
            _check(lib().ptts_voice_open(self.h, os.fsencode(src), C.byref(h)))
        return DeviceVoice(h.value, -1)

    def profile_enable(self, on):   # False / 0: off; True / 1: per-launch events + phases; 2: phases only
        _check(lib().ptts_profile_enable(self.h, int(on)))

    def profile_read(self) -> dict:
        p = _Profile()
        _check(lib().ptts_profile_read(self.h, C.byref(p)))
        return {"kernel": p.kernel.decode(), "launches": int(p.launches), "total_ms": float(p.total_ms),
                "algorithmic_bytes": float(p.algorithmic_bytes), "weight_bytes": float(p.weight_bytes),
                "prefill_ms": float(p.prefill_ms), "ar_loop_ms": float(p.ar_loop_ms), "mimi_ms": float(p.mimi_ms)}

    def new_batch(self, n_slots: int, kv_capacity: int) -> "Batch":
        return Batch(self, n_slots, kv_capacity)

    def decode_latents(self, latents, want_mimi_latent: bool = False):
        """LatentToMimi + MimiDecode (model.go:141,410): [n, frames, 32] -> pcm [n, frames*1920]."""
        lat = _f32(latents)
        if lat.ndim == 2:
            lat = lat[None]
        n, fr, _ = lat.shape
        pcm = np.empty((n, fr * self.info.samples_per_frame), np.float32)
        ml = np.empty((n, self.info.mimi_dim, fr), np.float32) if want_mimi_latent else None
        _check(lib().ptts_decode_latents(self.h, _fp(lat), n, fr, _fp(pcm), _fp(ml)))
        return (pcm, ml) if want_mimi_latent else pcm

    def decode_stages(self, latents):
        """decode_latents plus the decoder transformer's output rows [n, 16 frames, mimi_dim] (mimi.go:733-748):
        returns (pcm, mimi_latent, transformer_out)."""
        lat = _f32(latents)
        if lat.ndim == 2:
            lat = lat[None]
        n, fr, _ = lat.shape
        pcm = np.empty((n, fr * self.info.samples_per_frame), np.float32)
        ml = np.empty((n, self.info.mimi_dim, fr), np.float32)
        xf = np.empty((n, fr * self.info.steps_per_latent, self.info.mimi_dim), np.float32)
        _check(hooks().ptts_decode_stages(self.h, _fp(lat), n, fr, _fp(pcm), _fp(ml), _fp(xf)))
        return pcm, ml, xf

    def speaker_project(self, latent) -> np.ndarray:
        """projectSpeakerConditioning (onnx/voice_encode.go:119-158): Mimi-encoder latents [T, 512] -> voice embedding [T, d_model]."""
        lat = _f32(latent)
        lat = lat.reshape(-1, lat.shape[-1])
        out = np.empty((lat.shape[0], self.info.d_model), np.float32)
        L = lib()
        L.ptts_speaker_project.argtypes = [C.c_void_p, _FP, C.c_int64, _FP]
        _check(L.ptts_speaker_project(self.h, _fp(lat), lat.shape[0], _fp(out)))
        return out

    def noise_rows(self, seed: int, temperature: float, rows: int) -> np.ndarray:
        """The device draw of makeGaussianNoise (flow_lm.go:386-408) a request with (noise_seed, temperature) consumes."""
        out = np.empty((rows, self.info.ldim), np.float32)
        _check(lib().ptts_noise_rows(self.h, C.c_uint64(seed), C.c_float(temperature), rows, _fp(out)))
        return out

    def flow_direction(self, c, s: float, t: float, x) -> np.ndarray:
        c = _f32(c).reshape(-1, self.info.d_model)
        x = _f32(x).reshape(-1, self.info.ldim)
        out = np.empty_like(x)
        _check(lib().ptts_flow_direction(self.h, _fp(c), s, t, _fp(x), c.shape[0], _fp(out)))
        return out

    # -- batched GenerateAudio
    def _fill_request(self, r, toks, cfg, keep):
        t = toks if type(toks) is np.ndarray and toks.dtype == np.int64 and toks.flags.c_contiguous else np.ascontiguousarray(toks, np.int64)
        keep.append(t)
        r.tokens, r.n_tokens = _ip(t), t.size
        r.temperature, r.eos_threshold = cfg.temperature, min(cfg.eos_threshold, 3.0e38)
        r.max_steps, r.estimated_max_steps = cfg.max_steps, cfg.estimated_max_steps
        r.lsd_steps, r.frames_after_eos = cfg.lsd_decode_steps, cfg.frames_after_eos
        if cfg.voice_embedding is not None:
            ve = _f32(cfg.voice_embedding.data).reshape(-1, self.info.d_model)
            keep.append(ve)
            r.voice_embedding, r.voice_frames = _fp(ve), ve.shape[0]
        if cfg.voice_model_state is not None:
            ptrs, steps, offs, arrs = _voice_arrays(cfg.voice_model_state, self.info.n_layers)
            keep += [ptrs, steps, offs, arrs]
            r.voice_caches, r.voice_cache_steps, r.voice_offsets = ptrs, _ip(steps), _ip(offs)
        if cfg.device_voice is not None:
            r.voice = cfg.device_voice.h
        if cfg.noise is not None:
            nz = _f32(cfg.noise).reshape(-1, self.info.ldim)
            keep.append(nz)
            r.noise = _fp(nz)
            r.noise_rows = nz.shape[0]
        r.noise_seed = int(cfg.noise_seed)
        if cfg.step_callback is not None:
            cb = _STEP_CB(lambda _u, s, m, f=cfg.step_callback: f(s, m))
            keep.append(cb)
            r.step_callback = cb
        if cfg.cancel is not None:
            r.cancel = cfg.cancel.ctypes.data_as(C.POINTER(C.c_int32))
        r.want_latents = 1 if cfg.want_latents else 0
        r.pcm_format = 1 if cfg.pcm16 else 0
        if cfg.pcm_callback is not None:
            dt = np.int16 if r.pcm_format else np.float32

            def _pcm(_u, off, n, ptr, f=cfg.pcm_callback, dt=dt):
                f(int(off), np.frombuffer(C.string_at(ptr, int(n) * np.dtype(dt).itemsize), dtype=dt))

            pcb = _PCM_CB(_pcm)
            keep.append(pcb)
            r.pcm_callback = pcb
            r.stream_frames = int(cfg.stream_frames)

    def _take_result(self, rs, cfg) -> "GenerateResult":
        s16 = bool(getattr(cfg, "pcm16", False))   # PCM16 egress: int16 samples encoded on the device
        if rs.n_samples:
            # zero-copy: the array views the library's (page-locked) result buffer and gives it back to the pool
            # when it is garbage-collected
            pcm = np.asarray(_OwnedBuffer(rs.pcm16 if s16 else rs.pcm, int(rs.n_samples), "<i2" if s16 else "<f4"))
            if s16:
                rs.pcm16 = None
            else:
                rs.pcm = None
        else:
            pcm = np.zeros(0, np.int16 if s16 else np.float32)
        lat = None
        if cfg.want_latents:
            lat = np.ctypeslib.as_array(rs.latents, (rs.n_frames, self.info.ldim)).copy()
        return GenerateResult(pcm, int(rs.n_frames), int(rs.eos_step), lat)

    def generate_batch(self, token_lists: Sequence[Sequence[int]], cfgs: Sequence[RuntimeGenerateConfig]) -> list[GenerateResult]:
        n = len(token_lists)
        reqs = (_Request * n)()
        ress = (_Result * n)()
        keep = []
        for i, (toks, cfg) in enumerate(zip(token_lists, cfgs)):
            self._fill_request(reqs[i], toks, cfg, keep)
        rc = lib().ptts_generate(self.h, reqs, n, ress)
        out = []
        raw = np.frombuffer(ress, dtype=_RESULT_DTYPE, count=n)   # the result structs as one table: no per-field ctypes objects
        try:
            _check(rc)
            if raw["latents"].any():                                # want_latents: the general (slower) path
                for i in range(n):
                    out.append(self._take_result(ress[i], cfgs[i]))
            else:
                nf, es, ns = raw["n_frames"].tolist(), raw["eos_step"].tolist(), raw["n_samples"].tolist()
                a32, a16 = raw["pcm"].tolist(), raw["pcm16"].tolist()
                raw["pcm"][:] = 0                                   # ownership moves to the arrays below (zero-copy, freed on collection)
                raw["pcm16"][:] = 0
                for i in range(n):
                    s16 = a16[i] != 0
                    addr = a16[i] if s16 else a32[i]
                    if ns[i] and addr:
                        pcm = np.asarray(_OwnedBuffer(addr, ns[i], "<i2" if s16 else "<f4"))
                    else:
                        if addr:
                            _free_addr(addr)
                        pcm = np.zeros(0, np.int16 if cfgs[i].pcm16 else np.float32)
                    out.append(GenerateResult(pcm, nf[i], es[i], None))
        finally:
            if raw["pcm"].any() or raw["pcm16"].any() or raw["latents"].any():
                for i in range(n):
                    lib().ptts_free_result(C.byref(ress[i]))
        return out


class DeviceVoice:
    def __init__(self, handle: int, offset: int):
        self.h, self.offset = handle, offset

    def close(self):
        if self.h:
            lib().ptts_voice_free(self.h)
            self.h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


VOICE_FILE_UNKNOWN, VOICE_FILE_EMBEDDING, VOICE_FILE_MODEL_STATE = 0, 1, 2
VOICE_FILE_KIND_NAMES = {0: "unknown", 1: "embedding", 2: "model_state"}   # safetensors.VoiceFileKind (reader.go:20-26)


class VoiceFile:
    """A voice safetensors file read by the library (host only): InspectVoiceFile, LoadVoiceEmbedding, LoadVoiceModelState
    (internal/safetensors/reader.go:69-155) over ptts_voice_file_*."""

    def __init__(self, src):
        self.h = C.c_void_p()
        if isinstance(src, (bytes, bytearray, memoryview)):
            buf = bytes(src)
            _check(lib().ptts_voice_file_open_bytes(buf, len(buf), C.byref(self.h)))
        else:
            _check(lib().ptts_voice_file_open(os.fsencode(src), C.byref(self.h)))

    @property
    def kind(self) -> str:
        return VOICE_FILE_KIND_NAMES[int(lib().ptts_voice_file_kind(self.h))]

    def embedding(self) -> VoiceEmbedding:
        p, shape = _FP(), np.zeros(3, np.int64)
        _check(lib().ptts_voice_file_embedding(self.h, C.byref(p), _ip(shape)))
        n = int(shape.prod())
        data = np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.float32)
        return VoiceEmbedding(data.reshape(tuple(int(x) for x in shape)), [int(x) for x in shape])

    def model_state(self) -> VoiceModelState:
        n = C.c_int32()
        _check(lib().ptts_voice_file_modules(self.h, C.byref(n)))
        mods = {}
        for i in range(n.value):
            name, cache, off = C.c_char_p(), _VoiceTensor(), _VoiceTensor()
            _check(lib().ptts_voice_file_module(self.h, i, C.byref(name), C.byref(cache), C.byref(off)))
            m = {}
            for key, t in (("cache", cache), ("offset", off)):
                if not t.data:
                    continue
                shape = tuple(int(t.shape[d]) for d in range(t.rank))
                m[key] = (np.ctypeslib.as_array(t.data, shape=(int(t.count),)).copy() if t.count else np.zeros(0, np.float32)).reshape(shape)
            mods[name.value.decode()] = m
        return VoiceModelState(mods)

    def state_arrays(self, n_layers: int, heads: int = 0, head_dim: int = 0):
        """initStateFromVoiceModelState's checks (flow_transformer.go:451-590): per layer (cache [2,1,T,H,D], T, offset)."""
        ptrs, steps, offs = (_FP * max(n_layers, 1))(), np.zeros(max(n_layers, 1), np.int64), np.zeros(max(n_layers, 1), np.int64)
        _check(lib().ptts_voice_file_state(self.h, n_layers, heads, head_dim, ptrs, _ip(steps), _ip(offs)))
        return ptrs, steps[:n_layers], offs[:n_layers]

    def close(self):
        if self.h:
            lib().ptts_voice_file_close(self.h)
            self.h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def load_voice_conditioning(voice_path: str) -> dict:
    """tts.loadVoiceConditioning (service.go:216-246): {} for a blank path, else the keyword RuntimeGenerateConfig takes
    (voice_model_state or voice_embedding), chosen by the file's kind; error prefixes as in the reference."""
    if not voice_path or not voice_path.strip():
        return {}
    try:
        vf = VoiceFile(voice_path)
    except PttsError as e:
        raise PttsError(e.code, f"inspect voice safetensors: {e}") from None
    try:
        if vf.kind == "model_state":
            try:
                return {"voice_model_state": vf.model_state()}
            except PttsError as e:
                raise PttsError(e.code, f"load voice model state: {e}") from None
        try:
            return {"voice_embedding": vf.embedding()}
        except PttsError as e:
            raise PttsError(e.code, f"load voice embedding: {e}") from None
    finally:
        vf.close()


def _voice_arrays(state: VoiceModelState, n_layers: int):
    caches, steps, offs = [], [], []
    for i in range(n_layers):
        name = f"transformer.layers.{i}.self_attn"  # flow_transformer.go:513-515
        mod = state.modules.get(name)
        if mod is None:
            raise PttsError(PTTS_EINVAL, f'native: voice model state missing module "{name}"')
        if "cache" not in mod:
            raise PttsError(PTTS_EINVAL, f'native: voice model state module "{name}" missing cache')
        if "offset" not in mod:
            raise PttsError(PTTS_EINVAL, f'native: voice model state module "{name}" missing offset')
        c = _f32(mod["cache"])
        if c.ndim != 5 or c.shape[0] != 2:
            raise PttsError(PTTS_EINVAL, f'native: voice model state module "{name}" cache shape {list(c.shape)}, want [2,B,T,H,D]')
        if c.shape[1] != 1:
            raise PttsError(PTTS_EINVAL, f'native: voice model state module "{name}" batch {c.shape[1]}, want 1')
        if c.shape[3] != 16 or c.shape[4] != 64:
            raise PttsError(PTTS_EINVAL, f'native: voice model state module "{name}" heads {c.shape[3]}, want 16')
        off = np.asarray(mod["offset"], np.float32).reshape(-1)
        if off.size == 0:
            raise PttsError(PTTS_EINVAL, f'native: voice model state module "{name}" has empty offset tensor')
        if np.float32(int(off[0])) != off[0]:  # flow_transformer.go:554-566
            raise PttsError(PTTS_EINVAL, f'native: voice model state module "{name}" offset {off[0]} is not an integer')
        caches.append(c)
        steps.append(c.shape[2])
        offs.append(int(off[0]))
    ptrs = (_FP * n_layers)(*[_fp(c) for c in caches])
    return ptrs, np.array(steps, np.int64), np.array(offs, np.int64), caches


class Batch:
    """n_slots x native.FlowLMState (flow_lm.go:45-49): NewFlowState / PromptFlow / SampleNextLatentStateful."""

    def __init__(self, model: Model, n_slots: int, kv_capacity: int):
        self.model, self.n = model, n_slots
        h = C.c_void_p()
        _check(lib().ptts_batch_new(model.h, n_slots, kv_capacity, C.byref(h)))
        self.h = h.value

    def close(self):
        if self.h:
            lib().ptts_batch_free(self.h)
            self.h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():   # the HIP runtime may already be gone at interpreter teardown
            return
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def reset(self):
        _check(lib().ptts_batch_reset(self.h))

    def set_voice_state(self, slot: int, state: VoiceModelState):
        ptrs, steps, offs, arrs = _voice_arrays(state, self.model.info.n_layers)
        _check(lib().ptts_batch_set_voice_state(self.h, slot, ptrs, _ip(steps), _ip(offs)))

    def prompt(self, embs: Sequence[np.ndarray]):
        """embs[s]: [T_s, d_model] rows for slot s (voice rows first, then text rows)."""
        d = self.model.info.d_model
        rows = [_f32(e).reshape(-1, d) for e in embs]
        offs = np.zeros(self.n + 1, np.int64)
        offs[1:] = np.cumsum([r.shape[0] for r in rows])
        cat = np.concatenate(rows, 0) if offs[-1] else np.zeros((0, d), np.float32)
        cat = np.ascontiguousarray(cat)
        _check(lib().ptts_batch_prompt(self.h, _fp(cat), _ip(offs)))

    def step(self, frames_in, lsd_steps: int = 1, noise=None):
        m = self.model.info
        fi = _f32(frames_in).reshape(self.n, m.ldim)
        nz = _f32(noise).reshape(self.n, m.ldim) if noise is not None else None
        fo = np.empty((self.n, m.ldim), np.float32)
        eos = np.empty(self.n, np.float32)
        last = np.empty((self.n, m.d_model), np.float32)
        _check(lib().ptts_batch_step(self.h, _fp(fi), lsd_steps, _fp(nz), _fp(fo), _fp(eos), _fp(last)))
        return fo, eos, last

    def offsets(self) -> np.ndarray:
        o = np.zeros(self.n, np.int64)
        _check(lib().ptts_batch_offsets(self.h, _ip(o)))
        return o

    def read_kv(self, slot: int, layer: int):
        n = int(self.offsets()[slot])
        k = np.empty((self.model.info.n_heads, n, 64), np.float32)
        v = np.empty_like(k)
        _check(lib().ptts_batch_read_kv(self.h, slot, layer, _fp(k), _fp(v)))
        return k, v


class Runtime:
    """tts.Runtime (runtime.go:42-45) on one MI355X: GenerateAudio + Close (+ MimiTiming)."""

    def __init__(self, model: Model):
        self.model = model

    def generate_audio(self, tokens: Sequence[int], cfg: RuntimeGenerateConfig) -> np.ndarray:
        """GenerateAudio (runtime_native_safetensors.go:52-238): returns f32 PCM @ 24 kHz."""
        if self.model is None or self.model.h is None:
            raise PttsError(PTTS_EINVAL, "native-safetensors runtime unavailable")
        return self.model.generate_batch([tokens], [cfg])[0].pcm

    def generate(self, tokens: Sequence[int], cfg: RuntimeGenerateConfig) -> GenerateResult:
        if self.model is None or self.model.h is None:
            raise PttsError(PTTS_EINVAL, "native-safetensors runtime unavailable")
        return self.model.generate_batch([tokens], [cfg])[0]

    def mimi_timing(self) -> tuple[float, float, int]:
        """MimiTiming (runtime_native_safetensors.go:40-49): 12.5, 200, 16."""
        i = self.model.info
        return i.frame_rate, i.encoder_frame_rate, i.steps_per_latent

    def close(self):
        """Close (runtime_native_safetensors.go:240-244): nil-safe."""
        if self.model is not None:
            self.model.close()


# ----------------------------------------------------------------------------- kernel-level entry points

def op_linear(x, w, bias=None) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    out, inp = w.shape
    rows = x.size // inp
    y = np.empty(x.shape[:-1] + (out,), np.float32)
    b = _f32(bias) if bias is not None else None
    _check(lib().ptts_op_linear(_fp(x), _fp(w), _fp(b), rows, inp, out, _fp(y)))
    return y


def op_layernorm(x, w, b, eps: float) -> np.ndarray:
    x = _f32(x)
    d = x.shape[-1]
    y = np.empty_like(x)
    wa = _f32(w) if w is not None else None
    ba = _f32(b) if b is not None else None
    _check(lib().ptts_op_layernorm(_fp(x), _fp(wa), _fp(ba), eps, x.size // d, d, _fp(y)))
    return y


def op_rope(x, cos, sin, pos: int) -> np.ndarray:
    x = _f32(x).copy()
    cos, sin = _f32(cos), _f32(sin)
    seq, dim = x.shape[-2], x.shape[-1]
    _check(lib().ptts_op_rope(_fp(x), _fp(cos), _fp(sin), cos.shape[0], x.size // (seq * dim), seq, dim, pos))
    return x


def op_attention_positions(q, k, v, posq, posk, context: int) -> np.ndarray:
    q, k, v = _f32(q), _f32(k), _f32(v)
    b, h, tq, d = q.shape
    tk = k.shape[2]
    pq, pk = np.ascontiguousarray(posq, np.int64), np.ascontiguousarray(posk, np.int64)
    out = np.empty((b, h, tq, d), np.float32)
    _check(lib().ptts_op_attention_positions(_fp(q), _fp(k), _fp(v), b, h, tq, tk, d, _ip(pq), _ip(pk), context, _fp(out)))
    return out


def debug_tall_linear(x, w, *, bias=None, residual=None, epi=0, splitk=1, ln=None, planes=None, pbias=None, out_planes=False):
    """The AR step's 128+-row linear (csrc/tall.hip) on host operands (ptts_debug_tall_linear): `ln` = (weight, bias, eps) sends the rows through k_rowprep
    (with `planes` [psplit, M, K] and `pbias` summed in first); returns (out [splitk or 1, M, N] squeezed, updated rows or None)."""
    x, w = _f32(x), _f32(w)
    m, k = x.shape
    n = w.shape[0]
    s = max(1, int(splitk))
    out = np.empty((s, m, n), np.float32)
    lw = lb = None
    eps = 1e-5
    if ln is not None:
        lw, lb, eps = _f32(ln[0]), _f32(ln[1]), float(ln[2])
    pl = _f32(planes) if planes is not None else None
    pb = _f32(pbias) if pbias is not None else None
    b = _f32(bias) if bias is not None else None
    r = _f32(residual) if residual is not None else None
    xo = np.empty((m, k), np.float32) if (pl is not None and ln is not None) else None
    nul = C.cast(None, _FP)
    H = hooks()
    H.ptts_debug_tall_linear.argtypes = [C.c_int32] * 5 + [_FP, _FP, C.c_int32, _FP, _FP, _FP, C.c_float, _FP, _FP, _FP, C.c_int32, _FP, _FP]
    _check(H.ptts_debug_tall_linear(m, n, k, int(epi), s, _fp(x), _fp(pl) if pl is not None else nul, 0 if pl is None else pl.shape[0],
                                    _fp(pb) if pb is not None else nul, _fp(lw) if lw is not None else nul, _fp(lb) if lb is not None else nul, eps,
                                    _fp(w), _fp(b) if b is not None else nul, _fp(r) if r is not None else nul, 1 if out_planes else 0, _fp(out),
                                    _fp(xo) if xo is not None else nul))
    return (out[0] if s == 1 else out), xo


def last_attention_kernel() -> str:
    """Which kernel this thread's last attention launch used (ptts_debug_last_attention_kernel)."""
    return hooks().ptts_debug_last_attention_kernel().decode()


def launch_counts(on: bool) -> dict:
    """Kernel launches noted on this thread since the previous call ({kernel: count}); switches the census on / off."""
    buf = C.create_string_buffer(4096)
    hooks().ptts_debug_launch_counts(1 if on else 0, buf, 4096)
    return {k: int(v) for k, v in (item.split("=") for item in buf.value.decode().split(";") if item)}


def op_conv1d_leftpad(x, w, bias=None) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    b, cin, ln = x.shape
    cout, _, k = w.shape
    y = np.empty((b, cout, ln), np.float32)
    ba = _f32(bias) if bias is not None else None
    _check(lib().ptts_op_conv1d_leftpad(_fp(x), _fp(w), _fp(ba), b, cin, ln, cout, k, _fp(y)))
    return y


def op_convtr1d_righttrim(x, w, bias, stride: int, groups: int = 1) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    b, cin, ln = x.shape
    _, opg, k = w.shape
    y = np.empty((b, opg * groups, ln * stride), np.float32)
    ba = _f32(bias) if bias is not None else None
    _check(lib().ptts_op_convtr1d_righttrim(_fp(x), _fp(w), _fp(ba), b, cin, ln, opg, k, stride, groups, _fp(y)))
    return y


def wav_header_streaming() -> bytes:
    """audio.WriteWAVHeaderStreaming (internal/audio/wav_stream.go:15-41): the 44 bytes in front of a PCM16 stream."""
    buf = (C.c_uint8 * 44)()
    lib().ptts_wav_header_streaming(buf)
    return bytes(buf)


def op_pcm16(samples) -> np.ndarray:
    """audio.WritePCM16Samples on the device."""
    x = np.ascontiguousarray(samples, np.float32).reshape(-1)
    out = np.zeros(x.size, np.int16)
    L = lib()
    L.ptts_op_pcm16.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    _check(L.ptts_op_pcm16(x.ctypes.data, x.size, out.ctypes.data))
    return out


class _DispatchOpts(C.Structure):
    _fields_ = [("max_batch", C.c_int32), ("window_us", C.c_int32), ("queue_cap", C.c_int32), ("continuous", C.c_int32), ("cont_kv_capacity", C.c_int32),
                ("cont_max_steps", C.c_int32), ("cont_steps_per_group", C.c_int32), ("reserved", C.c_int32 * 1)]


class _DispatchStats(C.Structure):
    _fields_ = [("requests", C.c_int64), ("batches", C.c_int64), ("cancelled_waiting", C.c_int64), ("max_queue_depth", C.c_int64),
                ("mean_batch", C.c_double), ("mean_wait_us", C.c_double), ("mean_exec_us", C.c_double), ("cont_steps", C.c_int64), ("cont_slot_steps", C.c_int64), ("flow_cluster_fallbacks", C.c_int64)]


DISPATCH_EXEC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(_Request), C.c_int32, C.POINTER(_Result), C.c_void_p, C.c_int32)


class Dispatcher:
    """The serving front of the runtime: `generate` blocks like Synthesize behind the reference's worker
    semaphore (internal/server/server.go:398-421); concurrent callers are coalesced into batched GenerateAudio passes."""

    def __init__(self, models: Sequence[Model], max_batch: int = 0, window_us: int = 2000, queue_cap: int = 0, _custom_exec=None, _workers: int = 1,
                 continuous: Optional[bool] = None, cont_kv_capacity: int = 0, cont_max_steps: int = 0, cont_steps_per_group: int = 0):
        L = lib()
        L.ptts_dispatcher_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(_DispatchOpts), C.POINTER(C.c_void_p)]
        L.ptts_dispatcher_create_custom.argtypes = [DISPATCH_EXEC, C.c_void_p, C.c_int32, C.POINTER(_DispatchOpts), C.POINTER(C.c_void_p)]
        L.ptts_dispatch_generate.argtypes = [C.c_void_p, C.POINTER(_Request), C.POINTER(_Result)]
        L.ptts_dispatcher_stats.argtypes = [C.c_void_p, C.POINTER(_DispatchStats)]
        L.ptts_dispatcher_close.argtypes = [C.c_void_p]
        o = _DispatchOpts(max_batch=max_batch, window_us=window_us, queue_cap=queue_cap, continuous=0 if continuous is None else (1 if continuous else -1), cont_kv_capacity=cont_kv_capacity,
                          cont_max_steps=cont_max_steps, cont_steps_per_group=cont_steps_per_group)
        h = C.c_void_p()
        self.models = list(models)
        if _custom_exec is not None:
            self._exec = DISPATCH_EXEC(_custom_exec)   # kept alive with the dispatcher
            _check(L.ptts_dispatcher_create_custom(self._exec, None, _workers, C.byref(o), C.byref(h)))
        else:
            arr = (C.c_void_p * len(self.models))(*[m.h for m in self.models])
            _check(L.ptts_dispatcher_create(arr, len(self.models), C.byref(o), C.byref(h)))
        self.h = h.value

    def generate(self, tokens: Sequence[int], cfg: RuntimeGenerateConfig) -> GenerateResult:
        """Blocking; safe to call from many threads (the GIL is released while waiting)."""
        req, res, keep = _Request(), _Result(), []
        m = self.models[0] if self.models else None
        if m is not None:
            m._fill_request(req, tokens, cfg, keep)
        else:   # custom executor (tests): only what the queueing logic looks at
            t = np.ascontiguousarray(tokens, np.int64)
            keep.append(t)
            req.tokens, req.n_tokens = _ip(t), t.size
            if cfg.cancel is not None:
                req.cancel = cfg.cancel.ctypes.data_as(C.POINTER(C.c_int32))
        rc = lib().ptts_dispatch_generate(self.h, C.byref(req), C.byref(res))
        try:
            _check(rc)
            if m is None:
                return GenerateResult(np.zeros(0, np.float32), int(res.n_frames), int(res.eos_step), None)
            return m._take_result(res, cfg)
        finally:
            lib().ptts_free_result(C.byref(res))

    def stats(self) -> dict:
        s = _DispatchStats()
        lib().ptts_dispatcher_stats(self.h, C.byref(s))
        return {n: getattr(s, n) for n, _ in _DispatchStats._fields_ if n != "reserved"}

    def close(self):
        if self.h:
            lib().ptts_dispatcher_close(self.h)
            self.h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        self.close()


# ---- text front end (internal/text/prepare.go) ---------------------------------------------------------
def dsp_apply(samples, normalize: bool = False, dc_block: bool = False, fade_in_ms: float = 0.0, fade_out_ms: float = 0.0) -> np.ndarray:
    """audio.PeakNormalize / DCBlock / FadeIn / FadeOut in the CLI's order (dsp.go:12-78, synth.go:361-390); returns a new array."""
    out = np.array(samples, dtype=np.float32, copy=True).reshape(-1)
    L = lib()
    L.ptts_dsp_apply.argtypes = [_FP, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double]
    _check(L.ptts_dsp_apply(_fp(out), out.size, 1 if normalize else 0, 1 if dc_block else 0, float(fade_in_ms), float(fade_out_ms)))
    return out


def rccl_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 of a multi-GPU start-up)."""
    buf = (C.c_uint8 * 128)()
    _check(lib().ptts_rccl_unique_id(buf))
    return bytes(buf)


def rccl_broadcast(device_ptr: int, nbytes: int, rank: int, n_ranks: int, unique_id: bytes, device: int = 0):
    """The weight-arena broadcast (root = rank 0) over RCCL / xGMI, inside the library; every rank calls it."""
    L = lib()
    L.ptts_rccl_broadcast.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_char_p, C.c_int32]
    _check(L.ptts_rccl_broadcast(C.c_void_p(device_ptr), nbytes, rank, n_ranks, unique_id, device))


class Tokenizer:
    """tokenizer.Tokenizer (internal/tokenizer/tokenizer.go) backed by the library's SentencePiece unigram encoder
    (NewSentencePieceTokenizer / NewSentencePieceTokenizerFromBytes, sentencepiece.go:19-33)."""

    def __init__(self, model):
        L = lib()
        L.ptts_tokenizer_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.ptts_tokenizer_open_bytes.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.ptts_tokenizer_free.argtypes = [C.c_void_p]
        L.ptts_tokenizer_vocab_size.argtypes = [C.c_void_p]
        L.ptts_tokenizer_vocab_size.restype = C.c_int64
        L.ptts_tokenizer_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, _IP, C.c_int64]
        L.ptts_tokenizer_encode.restype = C.c_int64
        h = C.c_void_p()
        if isinstance(model, (bytes, bytearray)):
            buf = bytes(model)
            _check(L.ptts_tokenizer_open_bytes(buf, len(buf), C.byref(h)))
        else:
            _check(L.ptts_tokenizer_open(str(model).encode(), C.byref(h)))
        self.h = h.value

    @property
    def vocab_size(self) -> int:
        return int(lib().ptts_tokenizer_vocab_size(self.h))

    def encode(self, text: str) -> list:
        raw = text.encode("utf-8")
        ids = np.zeros(max(16, len(raw) + 2), np.int64)
        n = int(lib().ptts_tokenizer_encode(self.h, raw, len(raw), _ip(ids), ids.size))
        if n < 0:
            raise PttsError(PTTS_EINVAL, lib().ptts_last_error().decode())
        if n > ids.size:
            ids = np.zeros(n, np.int64)
            n = int(lib().ptts_tokenizer_encode(self.h, raw, len(raw), _ip(ids), ids.size))
        return [int(x) for x in ids[:n]]

    __call__ = encode

    def close(self):
        if self.h:
            lib().ptts_tokenizer_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def nfkc(text: str) -> str:
    """NFKC as the tokenizer applies it (ptts_text_nfkc: generated Unicode tables)."""
    raw = text.encode("utf-8")
    L = lib()
    L.ptts_text_nfkc.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]
    n = C.c_int64(0)
    buf = C.create_string_buffer(max(16, 20 * len(raw) + 16))   # one code point can expand to 18 (U+FDFA)
    _check(L.ptts_text_nfkc(raw, len(raw), buf, len(buf), C.byref(n)))
    return buf.raw[: n.value].decode("utf-8")


class _ChunkInfo(C.Structure):
    _fields_ = [("text", C.c_void_p), ("text_len", C.c_int64), ("token_ids", _IP), ("n_tokens", C.c_int64), ("num_words", C.c_int32),
                ("max_frames", C.c_int32), ("frames_after_eos", C.c_int32), ("reserved", C.c_int32)]


_ENCODE_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, _IP, C.c_int64)


@dataclass
class ChunkMetadata:
    """text.ChunkMetadata (prepare.go:18-24) + MaxFrames / FramesAfterEOS."""
    text: str
    token_ids: list
    num_words: int
    max_frames: int
    frames_after_eos: int

    @property
    def num_tokens(self) -> int:
        return len(self.token_ids)


def estimate_max_frames(token_count: int, frame_rate: float = 12.5) -> int:
    L = lib()
    L.ptts_text_estimate_max_frames.argtypes = [C.c_int64, C.c_double]
    return int(L.ptts_text_estimate_max_frames(token_count, frame_rate))


def frames_after_eos(num_words: int) -> int:
    L = lib()
    L.ptts_text_frames_after_eos.argtypes = [C.c_int64]
    return int(L.ptts_text_frames_after_eos(num_words))


def prepare_text(text: str) -> str:
    """text.PrepareText (prepare.go:66-100)."""
    L = lib()
    L.ptts_text_prepare.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    raw = text.encode("utf-8", "surrogatepass")
    n = C.c_int64(0)
    buf = C.create_string_buffer(len(raw) + 16)
    _check(L.ptts_text_prepare(raw, len(raw), buf, len(buf), C.byref(n)))
    return buf.raw[: n.value].decode("utf-8", "replace")


def prepare_chunks(text: str, encode: Callable[[str], Sequence[int]], max_tokens: int = 50, frame_rate: float = 12.5) -> list:
    """text.PrepareChunks (prepare.go:105-184); `encode` plays the Tokenizer interface (prepare.go:12-16)."""
    L = lib()
    L.ptts_text_chunks.argtypes = [C.c_char_p, C.c_int64, _ENCODE_FN, C.c_void_p, C.c_int32, C.c_double, C.POINTER(C.c_void_p)]
    L.ptts_chunks_count.argtypes = [C.c_void_p]
    L.ptts_chunks_get.argtypes = [C.c_void_p, C.c_int32, C.POINTER(_ChunkInfo)]
    L.ptts_chunks_free.argtypes = [C.c_void_p]

    def cb(_user, ptr, n, ids, cap):
        got = list(encode(C.string_at(ptr, n).decode("utf-8", "replace")))
        for i, v in enumerate(got[:cap]):
            ids[i] = int(v)
        return len(got)

    raw = text.encode("utf-8", "surrogatepass")
    h = C.c_void_p()
    if isinstance(encode, Tokenizer):   # the library's own encoder: no callback into Python (encode == NULL, user = tokenizer)
        _check(L.ptts_text_chunks(raw, len(raw), C.cast(None, _ENCODE_FN), encode.h, max_tokens, frame_rate, C.byref(h)))
    else:
        fn = _ENCODE_FN(cb)
        _check(L.ptts_text_chunks(raw, len(raw), fn, None, max_tokens, frame_rate, C.byref(h)))
    try:
        out = []
        for i in range(L.ptts_chunks_count(h)):
            ci = _ChunkInfo()
            _check(L.ptts_chunks_get(h, i, C.byref(ci)))
            out.append(ChunkMetadata(C.string_at(ci.text, ci.text_len).decode("utf-8", "replace"), [int(ci.token_ids[k]) for k in range(ci.n_tokens)],
                                     int(ci.num_words), int(ci.max_frames), int(ci.frames_after_eos)))
        return out
    finally:
        L.ptts_chunks_free(h)


# ---- tts.Service (internal/tts/service.go): text in, audio out --------------------------------------------------------------
MAX_TOKENS_PER_CHUNK = 50          # service.go:21-23
DEFAULT_MAX_STEPS = 256            # config.DefaultConfig().TTS.MaxSteps: "use the estimate" (service.go:271-278)


@dataclass
class TTSConfig:
    """The config.TTS fields generateConfig reads (service.go:255-269)."""
    temperature: float = 0.0
    eos_threshold: float = -4.0
    max_steps: int = DEFAULT_MAX_STEPS
    lsd_decode_steps: int = 1


class Service:
    """tts.Service: PrepareChunks -> one GenerateAudio per chunk -> concatenated PCM (service.go:107-153).

    The reference generates the chunks one after the other.  They are independent (each chunk starts from the voice
    state), so here ALL chunks of a text go to the runtime in one batched call (or through a Dispatcher, where they are
    coalesced with other callers' chunks): a one-minute text costs about what one 10-second chunk costs."""

    def __init__(self, model: Model, encode: Callable[[str], Sequence[int]], tts: Optional[TTSConfig] = None, dispatcher: Optional["Dispatcher"] = None):
        self.model, self.encode, self.tts, self.dispatcher = model, encode, tts or TTSConfig(), dispatcher

    def generate_config(self, chunk: "ChunkMetadata", **voice) -> RuntimeGenerateConfig:   # service.go:255-278
        frame_rate, _, spl = self.model.info.frame_rate, self.model.info.encoder_frame_rate, self.model.info.steps_per_latent
        est = estimate_max_frames(chunk.num_tokens, frame_rate)
        limit = est if est > 0 and (self.tts.max_steps <= 0 or self.tts.max_steps == DEFAULT_MAX_STEPS) else self.tts.max_steps
        return RuntimeGenerateConfig(temperature=self.tts.temperature, eos_threshold=self.tts.eos_threshold, max_steps=limit,
                                     estimated_max_steps=est, lsd_decode_steps=self.tts.lsd_decode_steps, frames_after_eos=chunk.frames_after_eos,
                                     mimi_steps_per_latent=spl, mimi_sequence_length=est * spl, **voice)

    def synthesize_chunks(self, text: str, **voice) -> list:
        """[(ChunkMetadata, GenerateResult)] in text order."""
        try:
            chunks = prepare_chunks(text, self.encode, MAX_TOKENS_PER_CHUNK, self.model.info.frame_rate)
        except PttsError as e:
            raise PttsError(e.code, f"no tokens produced from input: {e}") from None   # service.go:124-126
        cfgs = [self.generate_config(c, **voice) for c in chunks]
        if self.dispatcher is not None:
            import concurrent.futures as cf
            with cf.ThreadPoolExecutor(max_workers=len(chunks)) as ex:
                res = list(ex.map(lambda cc: self.dispatcher.generate(cc[0].token_ids, cc[1]), zip(chunks, cfgs)))
        else:
            res = self.model.generate_batch([c.token_ids for c in chunks], cfgs)
        return list(zip(chunks, res))

    def synthesize(self, text: str, **voice) -> np.ndarray:
        """Service.Synthesize (service.go:107-153): the chunks' PCM, concatenated."""
        parts = [r.pcm for _, r in self.synthesize_chunks(text, **voice)]
        return np.concatenate(parts) if parts else np.zeros(0, np.float32)
