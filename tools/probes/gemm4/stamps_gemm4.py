"""In-kernel clock stamps of the persistent k_gemm4 (block 0, first two tiles): python tools/stamps_gemm4.py [M N K]

per step: 0 top of step | 1 counted DMA wait done | 2 barrier passed | 3 DMA of the step two ahead issued | 4 MFMAs issued;
per tile: 5 K loop done | 6 epilogue stores issued.  Stamp 7 = s_memrealtime (100 MHz) at the top of the step -> the shader clock.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptts_amd

pkg = ptts_amd.load()
L = pkg.runtime.lib()
L.ptts_debug_gemm4_stamps.argtypes = [C.c_int32] * 3 + [C.POINTER(C.c_uint64)]
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32000, 1536, 512)
out = np.zeros((2, 64, 8, 8), np.uint64)
rc = L.ptts_debug_gemm4_stamps(M, N, K, out.ctypes.data_as(C.POINTER(C.c_uint64)))
if rc:
    raise SystemExit(L.ptts_last_error().decode())
ns = K // 32
t = out.astype(np.int64)
real = t[0, ns - 1, 0, 7] - t[0, 0, 0, 7]
ticks = t[0, ns - 1, 0, 0] - t[0, 0, 0, 0]
ghz = ticks / (real * 10.0) if real > 0 else float("nan")
print(f"k_gemm4 persistent, M={M} N={N} K={K}: {ns} steps per tile; shader clock over tile 0 = {ghz:.2f} GHz")
for ti in range(2):
    base = t[ti, 0, :, 0].min()
    print(f"tile {ti}: per step, median over the 8 waves (cycles): wait | barrier | dma issue | multiply | step total")
    tot = np.zeros(5)
    for s in range(ns):
        x = t[ti, s]
        d = [np.median(x[:, 1] - x[:, 0]), np.median(x[:, 2] - x[:, 1]), np.median(x[:, 3] - x[:, 2]), np.median(x[:, 4] - x[:, 3]), np.median(x[:, 4] - x[:, 0])]
        tot += d
        if s < 4 or s >= ns - 2:
            print(f"  step {s:2d}: {d[0]:7.0f} {d[1]:7.0f} {d[2]:7.0f} {d[3]:7.0f} {d[4]:7.0f}")
    print(f"  sum    : {tot[0]:7.0f} {tot[1]:7.0f} {tot[2]:7.0f} {tot[3]:7.0f} {tot[4]:7.0f}")
    x = t[ti, 63]
    kend = np.median(x[:, 5]); epi = np.median(x[:, 6])
    print(f"  K loop {np.median(x[:, 5] - t[ti, 0, :, 0]):.0f} cycles, epilogue {epi - kend:.0f} cycles")
