"""Back-to-back launch timing of the step linear: python tools/microbench.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptts_amd

pkg = ptts_amd.load()
L = pkg.runtime.hooks()   # the measurement entry points live in libptts_hooks.so (include/ptts_debug.h)
L.ptts_debug_time_skinny.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)]
shapes = [("in_proj", 3072, 1024), ("out_proj", 1024, 1024), ("linear1", 4096, 1024), ("linear2", 1024, 4096), ("flow 512x512", 512, 512),
          ("ada_all", 10240, 512), ("eos", 1, 1024), ("input_linear", 1024, 32)]
for bf16 in (1, 0):
    for M in (64, 1):
        for name, N, K in shapes:
            for S in ((1, 4) if K > 1024 else (1,)):
                if K // S > 1024:
                    continue
                for ln in ((0, 1) if (K <= 1024 and S == 1) else (0,)):
                    us = C.c_float(0)
                    rc = L.ptts_debug_time_skinny(M, N, K, bf16, S, ln, 200, C.byref(us))
                    if rc:
                        print(name, "unsupported", L.ptts_last_error().decode())
                        continue
                    wb = N * K * (2 if bf16 else 4)
                    print(f"{'bf16' if bf16 else 'f32 '} M={M:2d} {name:14s} N={N:5d} K={K:4d} S={S} ln={ln}: {us.value:7.2f} us  {wb/us.value/1e3:8.1f} GB/s (weights)")
