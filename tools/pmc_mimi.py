"""One batch-64 x 125-frame Mimi decode for counter collection:
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -d gpurun_out/pmc_mimi -o pmc --output-format csv -- python3 tools/pmc_mimi.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bench
import ptts_amd

pkg = ptts_amd.load()
wl = bench.WORKLOADS["b64_10s_bf16"]
path = bench.checkpoint_path(pkg, wl["file"], 0, lambda: None)
model = pkg.Model.open(path, device=0, weights=wl["weights"], kv=wl["kv"], max_batch=64)
lat = (np.random.default_rng(3).standard_normal((64, 125, 32)) * 0.5).astype(np.float32)
for _ in range(int(os.environ.get("PTTS_PMC_REPS", "1"))):
    pcm = model.decode_latents(lat)
print("decoded", pcm.shape, float(np.abs(pcm).max()))
