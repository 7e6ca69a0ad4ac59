"""Many-row GEMM kernels on the Mimi decoder's shapes (B=64, 125 frames): python tools/microbench_gemm.py [scale]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptts_amd

pkg = ptts_amd.load()
L = pkg.runtime.hooks()   # the measurement entry points live in libptts_hooks.so (include/ptts_debug.h)
L.ptts_debug_gemm.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 2
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25     # fraction of the B=64 row counts (host-side operand generation is slow)
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3]
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
R = 128000
# epilogue codes: +0x100 RoPE, +0x200 residual read from the output buffer, +0x400 prologue ELU (capi.cpp ptts_debug_gemm)
shapes = [("pf_qkv", 1600, 3072, 1024, 0x100), ("pf_out", 1600, 1024, 1024, 4), ("pf_l1", 1600, 4096, 1024, 1), ("pf_l2", 1600, 1024, 4096, 4), ("pf_l2_sk", 1600, 1024, 4096, 0x4000),
          ("qkv", R, 1536, 512, 0), ("qkv_rope", R, 1536, 512, 0x100), ("out_proj_ip", R, 512, 512, 0x205), ("out_proj_ls", R, 512, 512, 0xa05), ("ffn2_ip", R, 512, 2048, 0x205), ("rb1_0_elu", 6 * R, 128, 768, 0x403), ("rb2_0_re", 6 * R, 256, 128, 8), ("out_proj", R, 512, 512, 4), ("ffn1", R, 2048, 512, 1), ("ffn2", R, 512, 2048, 4),
          ("init_conv", R, 512, 3584, 3), ("up1", R, 1536, 1024, 0), ("rb1_0", 6 * R, 128, 768, 3), ("rb2_0", 6 * R, 256, 128, 4),
          ("up2", 6 * R, 640, 512, 0), ("rb1_1", 30 * R, 64, 384, 3), ("rb2_1", 30 * R, 128, 64, 4), ("up3", 30 * R, 256, 256, 0),
          ("rb1_2", 120 * R, 32, 192, 3), ("rb2_2", 120 * R, 64, 32, 4)]
for bf16 in (1,):
    for name, M, N, K, epi in shapes:
        if only and name not in only:
            continue
        if not name.startswith("pf_"):
            M = max(512, int(M * scale) // 256 * 256)
        if M * max(N, K) * 4 > 3e9:
            M = int(3e9 / (max(N, K) * 4)) // 256 * 256
        line = f"{name:10s} M={M:8d} N={N:5d} K={K:5d}"
        for v in variants:
            us, md = C.c_float(0), C.c_float(0)
            rc = L.ptts_debug_gemm(M, N, K, bf16, v, epi, 5, C.byref(us), C.byref(md))
            if rc:
                line += f" | v{v} unsupported: {L.ptts_last_error().decode()}"
                continue
            fl = 2.0 * M * N * K
            by = 4.0 * M * (K + N * (2 if (epi & 0xff) >= 4 else 1))
            line += f" | v{v} {us.value:9.1f} us {fl/us.value/1e6:7.1f} TF {by/us.value/1e3:7.0f} GB/s diff {md.value:.2e}"
        print(line, flush=True)
