// step_open.h -- the opening of an AR step on the matrix core: x = input_linear(previous frame, NaN -> bos) and
// fx = input_proj(x0) (flow_lm.go:247-255, flow_net.go:327), both ldim = 32 deep.
//
// Two kernels compute it: k_step_begin (kernels.hip: a launch of its own, in front of a group of steps) and the step's LAST launch
// (skinny.hip, k_skinny<..., FIN, CHAIN>: the block that has just produced the frame opens the next step as well, which removes one
// dependent launch per step).  Both go through the functions below on the same operand split, so a step opened either way starts
// from the same bits -- graph replay (chained inside a graph) and plain launches stay bit-identical.
#pragma once

#include "device_util.h"

namespace ptts {

typedef __bf16 so_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 so_bf16x2 __attribute__((ext_vector_type(2)));
typedef float so_f32x2 __attribute__((ext_vector_type(2)));
typedef float so_f32x4 __attribute__((ext_vector_type(4)));

constexpr int SO_K = 32;            // the latent width this path is built for (host: d.ldim == 32)
constexpr int SO_PITCH = SO_K * 2;  // bytes per row of a bf16 plane

union SoFrag {
    so_bf16x8 v;
    uint4 q;
    unsigned u[4];
};

// v -> bf16 hi + bf16 lo with v = hi + lo to ~2^-17 (round to nearest even; a NaN stays a NaN)
__device__ __forceinline__ void so_split1(float v, unsigned short& hi, unsigned short& lo) {
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}
__device__ __forceinline__ void so_split2(float a, float b, unsigned& hi, unsigned& lo) {
    so_f32x2 f = {a, b};
    so_bf16x2 h = __builtin_convertvector(f, so_bf16x2);
    so_f32x2 r = f - __builtin_convertvector(h, so_f32x2);
    so_bf16x2 l = __builtin_convertvector(r, so_bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

// element (row, col) of a [16][32] plane pair in LDS
__device__ __forceinline__ void so_put(unsigned char* ph, unsigned char* pl, int row, int col, float v) {
    unsigned short h, l;
    so_split1(v, h, l);
    *reinterpret_cast<unsigned short*>(ph + row * SO_PITCH + col * 2) = h;
    *reinterpret_cast<unsigned short*>(pl + row * SO_PITCH + col * 2) = l;
}

// the weight fragment (and bias piece) of one 16-column tile: lane (i = lane & 15, q = lane >> 4) holds W[n0 + i][8q .. 8q+7] and bias[n0 + 4q .. 4q+3];
// requested early (so_load_w: no dependence on the step's input), used by so_tile
template <bool WBF16> struct SoW {
    uint4 a;                        // bf16 weights: the eight; f32 weights: the first four floats
    uint4 b[WBF16 ? 0 : 1];         // f32 weights: the other four
    float4 bias;
};
template <bool WBF16>
__device__ __forceinline__ SoW<WBF16> so_load_w(const void* W, const float* bias, int n0, int N, int lane) {
    const int n = min(n0 + (lane & 15), N - 1), q = lane >> 4;
    SoW<WBF16> r;
    if constexpr (WBF16) {
        r.a = *reinterpret_cast<const uint4*>((const char*)W + (int64_t)n * (SO_K * 2) + q * 16);
    } else {
        const uint4* p = reinterpret_cast<const uint4*>((const char*)W + (int64_t)n * (SO_K * 4) + q * 32);
        r.a = p[0];
        r.b[0] = p[1];
    }
    // (no bias: any readable address, the value is discarded -- a load behind a condition would become a branch)
    const float4 bv = *reinterpret_cast<const float4*>((bias ? bias : reinterpret_cast<const float*>(W)) + min(n0 + 4 * q, N - 4));
    r.bias = bias ? bv : make_float4(0.f, 0.f, 0.f, 0.f);
    return r;
}

// out[m0 + j][n0 + i] = bias[n0 + i] + sum_k W[n0 + i][k] v[j][k] for 16 rows j and 16 columns i, v given as bf16 hi / lo planes in LDS.
// The product is computed transposed (weights as the matrix's rows), so a lane ends with four consecutive columns of one row:
// lane (j = lane & 15, q = lane >> 4) stores out[m0 + j][n0 + 4q .. 4q+3] as one 16-byte piece.  N % 16 == 0 (host).
template <bool WBF16>
__device__ __forceinline__ void so_tile(const unsigned char* ph, const unsigned char* pl, const SoW<WBF16>& w, int n0, float* out, int64_t ld, int m0, int M, int lane) {
    const int j = lane & 15, q = lane >> 4;
    SoFrag xh, xl;
    xh.q = *reinterpret_cast<const uint4*>(ph + j * SO_PITCH + q * 16);
    xl.q = *reinterpret_cast<const uint4*>(pl + j * SO_PITCH + q * 16);
    so_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (WBF16) {
        SoFrag wv;
        wv.q = w.a;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv.v, xh.v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv.v, xl.v, acc, 0, 0, 0);
    } else {
        SoFrag wh, wl;
        so_split2(__uint_as_float(w.a.x), __uint_as_float(w.a.y), wh.u[0], wl.u[0]);
        so_split2(__uint_as_float(w.a.z), __uint_as_float(w.a.w), wh.u[1], wl.u[1]);
        so_split2(__uint_as_float(w.b[0].x), __uint_as_float(w.b[0].y), wh.u[2], wl.u[2]);
        so_split2(__uint_as_float(w.b[0].z), __uint_as_float(w.b[0].w), wh.u[3], wl.u[3]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xh.v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xl.v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl.v, xh.v, acc, 0, 0, 0);
    }
    const int col = n0 + 4 * q;
    if (m0 + j < M) *reinterpret_cast<float4*>(out + (int64_t)(m0 + j) * ld + col) = make_float4(acc[0] + w.bias.x, acc[1] + w.bias.y, acc[2] + w.bias.z, acc[3] + w.bias.w);
}

}  // namespace ptts
