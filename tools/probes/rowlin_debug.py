"""Where does the fused norm1 + in_proj + RoPE kernel differ from the unfused path?  (debug aid; prints error maxima per 32-column chunk and per row)"""
import dataclasses, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
import ptts_amd
pkg = ptts_amd.load()
synth = pkg.synth
cfg = dataclasses.replace(synth.SynthConfig.tiny(), mimi_layers=2, mimi_ffn=2048, n_filters=16, layer_scale=1.0)
path = os.path.join(tempfile.mkdtemp(), "m.safetensors")
synth.write_safetensors(path, synth.make_checkpoint(cfg, seed=4242), dtype="BF16")
gm = pkg.Model.open(path, device=0, weights=1)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = (np.random.default_rng(1).standard_normal((rows, 512)) * 1.5 + 0.2).astype(np.float32)
got = gm.mimi_layer_qkv(0, x, 0, 0)
np.save(os.environ.get("OUT", "/tmp/rowlin_out.npy"), got)
if os.environ.get("REF"):
    ref = np.load(os.environ["REF"])
    err = np.abs(got - ref)
    print("max err", err.max(), "scale", np.abs(ref).max())
    print("per 32-col chunk:", np.round(err.reshape(rows, 48, 32).max(axis=(0, 2)), 4).tolist())
    print("per row (first 80):", np.round(err.max(axis=1)[:80], 4).tolist())
    print("within-chunk col pattern:", np.round(err.reshape(rows, 48, 32).max(axis=(0, 1)), 4).tolist())
