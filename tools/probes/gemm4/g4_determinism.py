"""Is the Mimi decoder transformer bit-reproducible under the current PTTS_GEMM4 setting?  python tools/g4_determinism.py
32 utterances x 64 frames (32768 rows) with the same latents in slots i and 31 - i, decoded three times: the transformer output
of slot i must equal that of slot 31 - i and that of the other runs, bit for bit."""
import dataclasses
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptts_amd

pkg = ptts_amd.load()
synth = pkg.synth
cfg = dataclasses.replace(synth.SynthConfig.tiny(), mimi_layers=2, mimi_ffn=2048, n_filters=16, layer_scale=1.0)
path = os.path.join(tempfile.mkdtemp(), "m.safetensors")
synth.write_safetensors(path, synth.make_checkpoint(cfg, seed=4242), dtype="BF16")
gm = pkg.Model.open(path, device=0, weights=1)
rng = np.random.default_rng(5)
half = (rng.standard_normal((16, 64, 32)) * 0.5).astype(np.float32)
x = np.concatenate([half, half[::-1]], 0)
pkg.runtime.launch_counts(True)
runs = [gm.decode_stages(x) for _ in range(3)]
print("PTTS_GEMM4 =", os.environ.get("PTTS_GEMM4"), pkg.runtime.launch_counts(False))
bad = 0
for r, (pcm, ml, xf) in enumerate(runs):
    for i in range(16):
        if not np.array_equal(xf[i], xf[31 - i]):
            d = np.abs(xf[i] - xf[31 - i])
            rows = np.nonzero(d.max(axis=1))[0]
            print(f"run {r}: slot {i} != slot {31 - i}: max diff {d.max():.3e} in {rows.size} rows, first rows {rows[:8]}, cols of first {np.nonzero(d[rows[0]])[0][:8]}")
            bad += 1
    if not np.array_equal(xf, runs[0][2]):
        d = np.abs(xf - runs[0][2])
        print(f"run {r} != run 0: max diff {d.max():.3e}, {np.count_nonzero(d.max(axis=(1, 2)))} slots differ")
        bad += 1
print("mismatches:", bad)
gm.close()
