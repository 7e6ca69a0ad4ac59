// skinny.hip -- the AR step's weight-streaming linear (the kernel the roofline in bench.py prices).
#include "kernels.h"
#include "device_util.h"

namespace ptts {

// ------------------------------------------------------------------------------------------------
// Weight-streaming linear for the AR step (K2-K4, K8, K9, K11 at M = batch <= 64 rows).
//
// The step is bound by streaming every weight once per step and, at these sizes (2-8 MB per
// matrix over 256 CUs), by latency: a block gets one shot at the memory system, and every instruction it executes
// is fetched cold.  So the kernel is one short burst per wave, with the work spread over many waves:
//   block  = 16 waves (1024 threads): 64 output columns x one 16-row tile of the batch x a K slice <= 1024;
//   wave w = column group (w & 3: 16 columns) x K quarter (w >> 2): it issues its <= 8 weight loads at once
//            (64 bf16 / 128 f32 contiguous bytes of one weight row per 128-deep super-step, coalesced 16-byte loads
//            straight to registers -- no LDS round trip for the operand that is read once) and stages ONE activation
//            row (wave w <-> row w of the tile).
// The activation tile is small and re-read by every column block (from L2); the prologue that stages it also does
// what would otherwise be separate launches on the critical path of the step:
//   * x += gate * (sum of the previous linear's split-K partials + bias)      (residual update, fixed order)
//   * LayerNorm (K3, linear.go:265-329), optionally without affine, optionally adaLN-modulated (K11)
// and writes the updated residual / normalised rows back once (column block 0 of each row tile).
// The row then lands in LDS as bf16 hi + lo halves (x = hi + lo to ~2^-17, v_cvt_pk_bf16_f32) in an XOR-swizzled
// image that the MFMA fragment reads hit conflict-free.  With bf16 weights the products are exact to the f32
// rounding of that split: two v_mfma_f32_16x16x32_bf16 per 32-deep step (hi*w, lo*w, separate accumulators); f32
// weights are split the same way in registers (hi*hi + lo*hi + hi*lo).  The k index inside a super-step is permuted
// identically for both operands (lane group q owns k = 32*q + 8*s + j in MFMA s), which makes the weight fetch
// contiguous.  The four K quarters of a column group are summed through LDS in a fixed order.
// Blocks that share a weight tile (the row tiles) are 8-congruent in dispatch order when N/64 is a multiple of 8,
// i.e. land on one XCD and share its L2 (speed only).
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SK_KMAX = 1024;   // K slice per block (LDS image: 2 x 16 rows x 2 KB = 64 KB)

// (a, b) -> packed bf16 hi pair and bf16 lo pair with a = hi + lo (round-to-nearest-even, NaN stays NaN)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);
    f32x2 r = f - __builtin_convertvector(h, f32x2);
    bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

union Frag8 {
    bf16x8 v;
    uint4 q;
    unsigned u[4];
};

template <bool WBF16>
__global__ __launch_bounds__(1024) void k_skinny(GemmArgs a, SkinnyFuse fu, int splitk, float* partial) {
    constexpr int WV = WBF16 ? 4 : 8;          // 16-byte weight loads per lane per super-step
    // LDS image: row r (0..15) = 2048 B = 128 chunks of 16 B; chunk c is stored at c ^ r, so the 16 lanes that read
    // the same logical chunk of 16 different rows hit 16 different bank groups
    __shared__ __attribute__((aligned(16))) unsigned char Xh[16 * 2048];
    __shared__ __attribute__((aligned(16))) unsigned char Xl[16 * 2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = wave & 3, kq4 = wave >> 2;
    const int n = blockIdx.x * 64 + cg * 16 + (lane & 15);
    const int m0 = blockIdx.y * 16, z = blockIdx.z, q = lane >> 4;
    const bool n_ok = n < a.N;
    const int kper = splitk > 1 ? ((a.K + splitk - 1) / splitk + 127) / 128 * 128 : a.K;
    const int k_begin = z * kper, k_end = min(a.K, k_begin + kper);
    const int klen = k_end - k_begin;          // <= SK_KMAX (host guarantees)
    const int nss = (klen + 127) >> 7;         // 128-deep super-steps in the slice (<= 8)
    const int ssq = (nss + 3) >> 2;            // super-steps per K quarter (<= 2)
    const int ss_lo = kq4 * ssq;

    // ---- weights: everything this wave will multiply is requested now ----
    uint4 w[2][WV];
    if (a.Wt) {   // fragment-ordered copy: one contiguous 1-KiB burst per wave-instruction
        const int nss_all = (a.K + 127) >> 7;
        const int tile = blockIdx.x * 4 + cg, ss_base = k_begin >> 7;
        const bool tile_ok = tile * 16 < a.N;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int ss = ss_lo + t;
            const uint4* src = reinterpret_cast<const uint4*>(a.Wt) + (((int64_t)tile * nss_all + ss_base + ss) * WV) * 64 + lane;
#pragma unroll
            for (int s = 0; s < WV; s++) {
                if (tile_ok && t < ssq && ss < nss) w[t][s] = src[s * 64];
                else w[t][s] = make_uint4(0, 0, 0, 0);
            }
        }
    } else {
        const char* wrow = (const char*)a.W + ((int64_t)(n_ok ? n : 0) * a.ldw + k_begin) * (WBF16 ? 2 : 4);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int kb = (ss_lo + t) * 128 + q * 32;   // this lane's 32 contiguous k of the super-step
#pragma unroll
            for (int s = 0; s < WV; s++) {
                const int k = kb + s * (WBF16 ? 8 : 4);
                if (n_ok && t < ssq && k < klen) w[t][s] = *reinterpret_cast<const uint4*>(wrow + (int64_t)k * (WBF16 ? 2 : 4));
                else w[t][s] = make_uint4(0, 0, 0, 0);
            }
        }
    }

    // ---- activations: wave w stages row w of the tile; lane owns float4 columns lane + 64 j ----
    {
        const int m = m0 + wave;
        const bool m_ok = m < a.M;
        float4 xr[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = (lane + 64 * j) * 4;
            xr[j] = (m_ok && k < klen) ? *reinterpret_cast<const float4*>(a.A + (int64_t)m * a.amap.ld + k_begin + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (fu.partial) {   // x += gate * (sum_z partial[z] + bias); the partials are added in a fixed order
            for (int zz = 0; zz < fu.psplit; zz++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = (lane + 64 * j) * 4;
                    if (m_ok && k < klen) {
                        float4 p = *reinterpret_cast<const float4*>(fu.partial + (int64_t)zz * fu.pstride + (int64_t)m * a.K + k);
                        if (zz == 0 && fu.pbias) { float4 b = *reinterpret_cast<const float4*>(fu.pbias + k); p.x += b.x; p.y += b.y; p.z += b.z; p.w += b.w; }
                        if (fu.pgate) { float4 g = *reinterpret_cast<const float4*>(fu.pgate + (int64_t)m * fu.ldpg + k); p.x *= g.x; p.y *= g.y; p.z *= g.z; p.w *= g.w; }
                        xr[j].x += p.x; xr[j].y += p.y; xr[j].z += p.z; xr[j].w += p.w;
                    }
                }
            }
            if (fu.x_out && blockIdx.x == 0 && m_ok) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = (lane + 64 * j) * 4;
                    if (k < klen) *reinterpret_cast<float4*>(fu.x_out + (int64_t)m * a.K + k) = xr[j];
                }
            }
        }
        if (fu.ln) {   // LayerNorm over the full row (K == row width), biased variance (linear.go:295-309)
            // the affine / modulation vectors are requested before the reductions, so their latency hides under them
            float4 lw[4], lb[4], lc[4], lh[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = (lane + 64 * j) * 4;
                const bool ok = k < klen;
                lw[j] = (ok && fu.ln_w) ? *reinterpret_cast<const float4*>(fu.ln_w + k) : make_float4(1.f, 1.f, 1.f, 1.f);
                lb[j] = (ok && fu.ln_b) ? *reinterpret_cast<const float4*>(fu.ln_b + k) : make_float4(0.f, 0.f, 0.f, 0.f);
                lc[j] = (ok && fu.scale && m_ok) ? *reinterpret_cast<const float4*>(fu.scale + (int64_t)m * fu.ldmod + k) : make_float4(0.f, 0.f, 0.f, 0.f);
                lh[j] = (ok && fu.scale && m_ok) ? *reinterpret_cast<const float4*>(fu.shift + (int64_t)m * fu.ldmod + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) s += (xr[j].x + xr[j].y) + (xr[j].z + xr[j].w);
            const float mean = wave_sum(s) / (float)a.K;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if ((lane + 64 * j) * 4 < klen) {
                    float d0 = xr[j].x - mean, d1 = xr[j].y - mean, d2 = xr[j].z - mean, d3 = xr[j].w - mean;
                    v += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            }
            const float inv_std = 1.0f / sqrtf(wave_sum(v) / (float)a.K + fu.eps);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = (lane + 64 * j) * 4;
                if (k >= klen) continue;
                float4 o;
                o.x = ((xr[j].x - mean) * inv_std * lw[j].x + lb[j].x) * (lc[j].x + 1.0f) + lh[j].x;
                o.y = ((xr[j].y - mean) * inv_std * lw[j].y + lb[j].y) * (lc[j].y + 1.0f) + lh[j].y;
                o.z = ((xr[j].z - mean) * inv_std * lw[j].z + lb[j].z) * (lc[j].z + 1.0f) + lh[j].z;
                o.w = ((xr[j].w - mean) * inv_std * lw[j].w + lb[j].w) * (lc[j].w + 1.0f) + lh[j].w;
                xr[j] = o;
                if (fu.y_out && blockIdx.x == 0 && m_ok) *reinterpret_cast<float4*>(fu.y_out + (int64_t)m * a.K + k) = o;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = (lane + 64 * j) * 4;
            if (k >= nss * 128) continue;
            unsigned h01, l01, h23, l23;
            split2(xr[j].x, xr[j].y, h01, l01);
            split2(xr[j].z, xr[j].w, h23, l23);
            const int off = wave * 2048 + ((((k >> 3) ^ wave) & 127) << 4) + ((k & 4) << 1);
            *reinterpret_cast<uint2*>(&Xh[off]) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(&Xl[off]) = make_uint2(l01, l23);
        }
    }
    __syncthreads();

    f32x4 acc_h = {0.f, 0.f, 0.f, 0.f}, acc_l = {0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        if (t >= ssq || ss_lo + t >= nss) break;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int c = (ss_lo + t) * 16 + q * 4 + s;            // logical 16-byte chunk = 8 k
            const int off = i16 * 2048 + (((c ^ i16) & 127) << 4);
            Frag8 xh, xl;
            xh.q = *reinterpret_cast<const uint4*>(&Xh[off]);
            xl.q = *reinterpret_cast<const uint4*>(&Xl[off]);
            if constexpr (WBF16) {
                Frag8 wv;
                wv.q = w[t][s];
                acc_h = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh.v, wv.v, acc_h, 0, 0, 0);
                acc_l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl.v, wv.v, acc_l, 0, 0, 0);
            } else {
                Frag8 wh, wl;
                const uint4 r0 = w[t][2 * s], r1 = w[t][2 * s + 1];
                split2(__uint_as_float(r0.x), __uint_as_float(r0.y), wh.u[0], wl.u[0]);
                split2(__uint_as_float(r0.z), __uint_as_float(r0.w), wh.u[1], wl.u[1]);
                split2(__uint_as_float(r1.x), __uint_as_float(r1.y), wh.u[2], wl.u[2]);
                split2(__uint_as_float(r1.z), __uint_as_float(r1.w), wh.u[3], wl.u[3]);
                acc_h = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh.v, wh.v, acc_h, 0, 0, 0);
                acc_l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl.v, wh.v, acc_l, 0, 0, 0);
                acc_l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh.v, wl.v, acc_l, 0, 0, 0);
            }
        }
    }
    // ---- sum the four K quarters of each column group (fixed order), reusing the activation image ----
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(Xh);   // [kq4][cg][lane]
    const f32x4 accv = acc_h + acc_l;
    if (kq4 > 0) red[(kq4 * 4 + cg) * 64 + lane] = make_float4(accv[0], accv[1], accv[2], accv[3]);
    __syncthreads();
    if (kq4 > 0 || !n_ok) return;
    float acc[4] = {accv[0], accv[1], accv[2], accv[3]};
#pragma unroll
    for (int t = 1; t < 4; t++) {
        float4 p = red[(t * 4 + cg) * 64 + lane];
        acc[0] += p.x; acc[1] += p.y; acc[2] += p.z; acc[3] += p.w;
    }
    // D layout of 16x16x32: column (n) = lane & 15, row (m) = (lane >> 4) * 4 + reg
    if (splitk > 1) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            int m = m0 + q * 4 + reg;
            if (m < a.M) partial[((int64_t)z * a.M + m) * a.N + n] = acc[reg];
        }
        return;
    }
    const float bias = a.bias ? a.bias[n] : 0.0f;
    const float addv = a.addvec ? a.addvec[n] : 0.0f;
    const float scl = a.scale ? a.scale[n] : 1.0f;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        int m = m0 + q * 4 + reg;
        if (m >= a.M) continue;
        float v = acc[reg] + bias;
        int64_t co = (int64_t)m * a.cmap.ld + n;
        switch (a.epi) {
            case EPI_NONE: break;
            case EPI_GELU: v = gelu1(v); break;
            case EPI_SILU: v = silu1(addv + v); break;
            case EPI_ELU: v = elu1(v); break;
            case EPI_RESADD: v = a.R[co] + v; break;
            case EPI_SCALE_RESADD: v = a.R[co] + scl * v; break;
            case EPI_GATE_RESADD: v = a.R[co] + a.gate[(int64_t)m * a.ldg + n] * v; break;
            case EPI_AXPY: v = a.R[co] + a.alpha * v; break;
        }
        a.C[co] = v;
    }
}

bool skinny_supported(const GemmArgs& a, int splitk) {
    if (splitk < 1) splitk = 1;
    const int kslice = splitk > 1 ? ((a.K + splitk - 1) / splitk + 127) / 128 * 128 : a.K;
    return a.M <= 64 && a.K % 8 == 0 && kslice <= SK_KMAX && a.amap.rows_per_batch == 0 && a.cmap.rows_per_batch == 0 &&
           a.amap.ld % 4 == 0 && aligned16(a.A) && aligned16(a.W) && a.ldw % 8 == 0 && a.aop == AOP_NONE;
}

bool skinny_fuse_supported(const GemmArgs& a, const SkinnyFuse& f) {
    // the fused prologue needs whole rows in one block: K is the row width, one K slice, dense rows
    return skinny_supported(a, 1) && a.K <= SK_KMAX && a.amap.ld == a.K && a.K % 4 == 0 && (!f.scale || f.ldmod % 4 == 0) &&
           (!f.pgate || f.ldpg % 4 == 0);
}

void launch_skinny(const GemmArgs& a, const SkinnyFuse& fu, int splitk, float* partial, hipStream_t stream) {
    if (a.M <= 0 || a.N <= 0) return;
    dim3 grid((a.N + 63) / 64, (a.M + 15) / 16, splitk);
    if (a.w_bf16) hipLaunchKernelGGL(k_skinny<true>, grid, dim3(1024), 0, stream, a, fu, splitk, partial);
    else hipLaunchKernelGGL(k_skinny<false>, grid, dim3(1024), 0, stream, a, fu, splitk, partial);
}

}  // namespace ptts
