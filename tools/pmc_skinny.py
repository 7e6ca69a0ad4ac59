"""A few launches of the step linear for counter collection: rocprofv3 --pmc ... -- python3 tools/pmc_skinny.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptts_amd

pkg = ptts_amd.load()
L = pkg.runtime.hooks()   # the measurement entry points live in libptts_hooks.so (include/ptts_debug.h)
L.ptts_debug_time_skinny.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)]
for name, N, K, S, ln in [("eos", 1, 1024, 1, 0), ("flow512", 512, 512, 1, 1), ("qkv", 3072, 1024, 1, 1), ("ffn2", 1024, 4096, 4, 0)]:
    us = C.c_float(0)
    L.ptts_debug_time_skinny(64, N, K, 1, S, ln, 20, C.byref(us))
    print(name, us.value)
