"""Transformer output of a 16 x 64-frame batch (16384 rows, bf16 weights) saved for comparison between builds / environment settings:
    PTTS_SPLIT_ROWS=0 python tools/probes/split_rows/dump_xf.py out0.npy ; PTTS_SPLIT_ROWS=1 python ... out1.npy ; python ... --cmp out0.npy out1.npy"""
import dataclasses, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a - b)
    print("max abs diff", d.max(), "differing values", int((d != 0).sum()), "of", d.size)
    rows = np.where(d.reshape(-1, d.shape[-1]).max(axis=1) != 0)[0]
    print("rows differing", rows.size, "first", rows[:20])
    cols = np.where(d.reshape(-1, d.shape[-1]).max(axis=0) != 0)[0]
    print("cols differing", cols.size, "first", cols[:40])
    sys.exit(0)
import ptts_amd
pkg = ptts_amd.load()
synth = pkg.synth
cfg = dataclasses.replace(synth.SynthConfig.tiny(), mimi_layers=int(os.environ.get("LAYERS", "2")), mimi_ffn=2048, n_filters=16, layer_scale=1.0)
path = "/tmp/mimi_full_BF16.safetensors"
synth.write_safetensors(path, synth.make_checkpoint(cfg, seed=4242), dtype="BF16")
gm = pkg.Model.open(path, device=0, weights=1)
rng = np.random.default_rng(11)
x = (rng.standard_normal((16, 64, 32)) * 0.5).astype(np.float32)
_, _, xf = gm.decode_stages(x)
np.save(sys.argv[1], xf)
gm.close()
