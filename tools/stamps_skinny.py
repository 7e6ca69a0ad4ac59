"""Phase timeline of the step linear from in-kernel clock stamps: python tools/stamps_skinny.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptts_amd

pkg = ptts_amd.load()
L = pkg.runtime.hooks()   # the measurement entry points live in libptts_hooks.so (include/ptts_debug.h)
L.ptts_debug_skinny_stamps.argtypes = [C.c_int32] * 6 + [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
names = ["entry", "w issued", "x+LN done", "staged+sync", "mfma done", "k-reduce", "stored"]
for name, N, K, S, ln in [("eos", 1, 1024, 1, 0), ("flow512", 512, 512, 1, 1), ("out_proj", 1024, 1024, 1, 0), ("qkv", 3072, 1024, 1, 1),
                          ("ffn1", 4096, 1024, 1, 1), ("ffn2", 1024, 4096, 4, 0)]:
    buf = np.zeros((4096, 8), dtype=np.uint64)
    nb = C.c_int32(0)
    rc = L.ptts_debug_skinny_stamps(64, N, K, 1, S, ln, buf.ctypes.data, 4096, C.byref(nb))
    if rc:
        print(name, L.ptts_last_error().decode())
        continue
    b = buf[: nb.value].astype(np.int64)
    real = b[:, 7]
    t0 = real.min()
    rel = (b[:, 1:7] - b[:, 0:1])          # ticks since block entry
    print(f"{name:9s} blocks={nb.value:4d}  entry spread (100 MHz wall): first {0} last {(real.max()-t0)*10} ns")
    for q in (0, 50, 100):
        r = np.percentile(rel, q, axis=0)
        print(f"   p{q:<3d} " + "  ".join(f"{n}:{int(v):6d}" for n, v in zip(names[1:], r)))
