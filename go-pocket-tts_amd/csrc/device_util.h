// device_util.h -- small device/host helpers shared by the .hip files.
#pragma once

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

#include "kernels.h"

// The lo half of every activation split in the DECODER's products (x = hi + lo: two MFMAs per product against bf16 weights).  -DPTTS_ABLATE_LO drops them: the
// measurement build of tools/probes/hi_only ("what do the f32-grade sums cost on a configuration that says bf16": its time and its error against the oracle are in
// profiles/r5_hi_only_ablation.txt).  Nothing in the shipped library defines it.
#ifdef PTTS_ABLATE_LO
#define PTTS_LO_MFMA(stmt) do { } while (0)
#else
#define PTTS_LO_MFMA(stmt) stmt
#endif

namespace ptts {

#define WAVE 64

__device__ __forceinline__ int64_t row_off(const RowMap& m, int64_t r) {
    if (!m.rows_per_batch) return r * m.ld;
    const unsigned rpb = (unsigned)m.rows_per_batch, ru = (unsigned)r;   // row counts fit in 32 bits: avoid the 64-bit divide
    const unsigned b = ru / rpb, t = ru - b * rpb;
    return (int64_t)b * m.batch_stride + (int64_t)t * m.ld;
}

__device__ __forceinline__ float elu1(float v) { return v <= 0.0f ? expf(v) - 1.0f : v; }        // tensor_util.go:119-128
__device__ __forceinline__ float silu1(float v) { return v / (1.0f + expf(-v)); }                  // tensor_util.go:73-82
// ELU on the hardware exponential (v_exp_f32 of v*log2(e)): |error| < 2e-7 absolute on (-inf, 0], below the rounding noise of the
// bf16-split products it feeds; used where ELU sits in a GEMM prologue/epilogue and is evaluated millions of times per launch
__device__ __forceinline__ float elu_fast(float v) { return v <= 0.0f ? __expf(v) - 1.0f : v; }
__device__ __forceinline__ float gelu1(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }  // :84-94

// audio.WritePCM16Samples (audio/wav_stream.go:43-54) for one sample
__device__ __forceinline__ int pcm16_one(float s) {
    double c = (double)s;                       // the reference clamps and multiplies in float64: the product is exact
    c = c > 1.0 ? 1.0 : c;
    c = c < -1.0 ? -1.0 : c;
    return s != s ? 0 : (int)(c * 32767.0);     // float -> int conversion truncates toward zero, like Go's int16(x)
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
// 64-lane float sum on the DPP path (quad swaps, row mirrors, then four v_readlane): ~10 issue slots instead of six
// ds_bpermute round trips (~100 cycles each) -- it sits on the critical path of the fused LayerNorm prologue.  Fixed order.
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);   // row_half_mirror
    v += dpp_f32<0x140>(v);   // row_mirror: every lane now holds the sum of its row of 16
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace ptts
