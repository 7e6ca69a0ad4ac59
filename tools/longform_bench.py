"""Long-form synthesis (BASELINE configs[4] without cloning / int8): a one-minute text through tts.Service.

The reference generates the chunks of a text one after the other; here they are one batch.  python tools/longform_bench.py"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bench
import ptts_amd

pkg = ptts_amd.load()
cfg = pkg.synth.SynthConfig.full()
wl = bench.WORKLOADS["b64_10s_bf16"]
path = bench.checkpoint_path(pkg, wl["file"], 0, lambda: None)
model = pkg.Model.open(path, device=0, weights=wl["weights"], kv=wl["kv"], max_batch=64)
voice = model.upload_voice(pkg.VoiceModelState(bench.voice_modules(pkg, cfg)))


def encode(t):   # stands in for SentencePiece (no tokenizer.model offline): ~1.3 tokens per word
    out = []
    for w in t.split():
        h = sum(ord(c) * (k + 1) for k, c in enumerate(w))
        out.append(1 + h % 3999)
        if len(w) > 6:
            out.append(1 + (h * 7) % 3999)
    return out


sentence = "the quick brown fox jumps over the lazy dog while the old clock in the hall strikes thirteen and nobody in the house seems to mind at all."
for n_sent in (1, 6, 24):
    text = " ".join([sentence] * n_sent)
    svc = pkg.Service(model, encode, pkg.TTSConfig(eos_threshold=float("inf"), max_steps=125))   # synthetic weights never say EOS: 10 s per chunk
    times = []
    for _ in range(4):
        t0 = time.perf_counter()
        pairs = svc.synthesize_chunks(text, device_voice=voice)
        times.append(time.perf_counter() - t0)
    audio = sum(r.n_frames for _, r in pairs) * bench.FRAME_SEC
    dt = statistics.median(times[1:])
    print(f"{n_sent:2d} sentences -> {len(pairs):2d} chunks, {audio:6.1f} s of audio in {1e3*dt:7.1f} ms = {audio/dt:7.0f} x real time "
          f"(one chunk after the other, as the reference does: about {len(pairs)} x {1e3*dt/max(1,len(pairs)) if len(pairs)==1 else 28.0:.0f} ms)", flush=True)
voice.close()
model.close()
