"""Race check of the many-row GEMM kernels: the same product N times, bits compared.  python tools/g4_repeat.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptts_amd

pkg = ptts_amd.load()
L = pkg.runtime.lib()
L.ptts_debug_gemm_repeat.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_int32), C.POINTER(C.c_float)]
for shape in [(32768, 1536, 512, 0)]:
    for mode in (1,):
        for v in (51,):
            bad, md = C.c_int32(0), C.c_float(0)
            rc = L.ptts_debug_gemm_repeat(*shape[:3], v, shape[3], 8, mode, C.byref(bad), C.byref(md))
            print(shape, "mode", mode, "variant", v, "rc", rc, "runs that differ from the first:", bad.value, "max diff", md.value, flush=True)
