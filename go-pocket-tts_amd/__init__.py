"""MI355X-native PocketTTS synthesis path (drop-in for go-pocket-tts' tts.Runtime seam).

csrc/      hand-written HIP kernels for gfx950 + the C++ engine behind the C ABI (include/ptts.h)
runtime.py host-side mirror of tts.Runtime / native.Model over that C ABI
synth.py   synthetic checkpoints / voices / prompts (no real weights exist offline)
"""
from . import runtime, synth  # noqa: F401
from .runtime import (Batch, Cancelled, DeviceVoice, Dispatcher, GenerateResult, Model, PttsError, Runtime, RuntimeGenerateConfig, Service, TTSConfig,  # noqa: F401
                      VoiceEmbedding, VoiceFile, VoiceModelState, load_voice_conditioning, KV_BF16, KV_F32, WEIGHTS_BF16, WEIGHTS_F32, WEIGHTS_INT8)
