// gemm2.hip -- the many-row GEMM behind prefill and the Mimi decoder (every Linear / Conv1d / ConvTranspose1d of
// mimi.go:719-789 and flow_transformer.go:749-771 at M = hundreds .. millions of rows).
#include "kernels.h"
#include "device_util.h"

namespace ptts {

// C[M,N] = epi( aop(A)[M,K] * W[N,K]^T ) on the bf16 matrix cores with f32-grade accuracy:
// activations are f32 in HBM and are split while they are staged into LDS into bf16 hi + lo halves
// (x = hi + lo to ~2^-17, v_cvt_pk_bf16_f32); bf16 weights multiply both halves (2 x v_mfma_f32_32x32x16_bf16),
// f32 weights are split the same way and take three products (hi*hi + lo*hi + hi*lo; the dropped lo*lo term is
// ~2^-18 relative).  Accumulation is f32 in the MFMA accumulators.  Against the exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate) this is 5-8x the arithmetic rate at the same tolerance class.
//
// Tile: 128 rows x BN columns (BN = 128 / 64 / 32 picked from N) x 32-deep k tiles, 4 waves, each wave 32 rows x BN.
// LDS rows are 32 bf16 + 8 pad (80 B): the 16-lane groups of a ds_read_b128 then start 20 dwords apart, which is
// conflict-free.  Global loads of tile t+1 are issued before the MFMAs of tile t (register prefetch).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split2g(float a, float b, unsigned& hi, unsigned& lo) {
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);
    f32x2 r = f - __builtin_convertvector(h, f32x2);
    bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

union FragG {
    bf16x8 v;
    uint4 q;
};

constexpr int G2_BM = 128, G2_BK = 32, G2_LD = 40;   // LD in bf16 elements (80-byte rows)

template <int BN, bool WBF16>
__global__ __launch_bounds__(256) void k_gemm2(GemmArgs a) {
    constexpr int NT = BN / 32;                         // 32-column MFMA tiles per wave
    constexpr int WCH = WBF16 ? (BN * 4 + 255) / 256 : (BN * 8 + 255) / 256;   // 16-byte weight chunks per thread per k tile
    __shared__ __attribute__((aligned(16))) unsigned short Ah[G2_BM * G2_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Al[G2_BM * G2_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Wh[BN * G2_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Wl[WBF16 ? 8 : BN * G2_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order: blocks are dealt round-robin to the 8 XCDs (each with a private L2), so block id b runs on
    // XCD b % 8.  All column tiles of one 128-row panel of A are given ids with the same b % 8 and consecutive b / 8:
    // the panel is fetched from HBM once into that XCD's L2 instead of once per column tile.  (Speed only.)
    const int ncol = (a.N + BN - 1) / BN, npan = (a.M + G2_BM - 1) / G2_BM;
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int pan = (j / ncol) * 8 + xcd;
    if (pan >= npan) return;
    const int m0 = pan * G2_BM, n0 = (j % ncol) * BN;

    // A staging: thread owns rows (tid >> 3) + 32 i, float4 column (tid & 7)
    const float* aptr[4];
    bool aok[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int gm = m0 + (tid >> 3) + 32 * i;
        aok[i] = gm < a.M;
        aptr[i] = a.A + (aok[i] ? row_off(a.amap, gm) : 0) + (tid & 7) * 4;
    }
    // W staging: chunk c = tid + 256 j; bf16: row = c >> 2, 8 k at (c & 3) * 8; f32: row = c >> 3, 4 k at (c & 7) * 4
    const char* wptr[WCH];
    bool wok[WCH];
    int wrow[WCH], wk[WCH];
#pragma unroll
    for (int j = 0; j < WCH; j++) {
        int c = tid + 256 * j;
        wrow[j] = WBF16 ? c >> 2 : c >> 3;
        wk[j] = WBF16 ? (c & 3) * 8 : (c & 7) * 4;
        wok[j] = wrow[j] < BN && n0 + wrow[j] < a.N;
        wptr[j] = (const char*)a.W + ((int64_t)(wok[j] ? n0 + wrow[j] : 0) * a.ldw + wk[j]) * (WBF16 ? 2 : 4);
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.0f;

    float4 ar[4];
    uint4 wr[WCH];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int k = k0 + (tid & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (aok[i]) {
                if (k + 3 < a.K) v = *reinterpret_cast<const float4*>(aptr[i] + k0);
                else {
                    if (k + 0 < a.K) v.x = aptr[i][k0 + 0];
                    if (k + 1 < a.K) v.y = aptr[i][k0 + 1];
                    if (k + 2 < a.K) v.z = aptr[i][k0 + 2];
                }
                if (a.aop == AOP_ELU) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
            }
            ar[i] = v;
        }
#pragma unroll
        for (int j = 0; j < WCH; j++) {
            int k = k0 + wk[j];
            uint4 u = make_uint4(0, 0, 0, 0);
            if (wok[j] && k < a.K) u = *reinterpret_cast<const uint4*>(wptr[j] + (int64_t)k0 * (WBF16 ? 2 : 4));   // K % 8 == 0 (bf16) / % 4 (f32)
            wr[j] = u;
        }
    };
    gload(0);
    for (int k0 = 0; k0 < a.K; k0 += G2_BK) {
        __syncthreads();   // the previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < 4; i++) {
            unsigned h01, l01, h23, l23;
            split2g(ar[i].x, ar[i].y, h01, l01);
            split2g(ar[i].z, ar[i].w, h23, l23);
            int off = ((tid >> 3) + 32 * i) * G2_LD + (tid & 7) * 4;
            *reinterpret_cast<uint2*>(&Ah[off]) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(&Al[off]) = make_uint2(l01, l23);
        }
#pragma unroll
        for (int j = 0; j < WCH; j++) {
            if (wrow[j] >= BN) continue;
            if constexpr (WBF16) {
                *reinterpret_cast<uint4*>(&Wh[wrow[j] * G2_LD + wk[j]]) = wr[j];
            } else {
                unsigned h01, l01, h23, l23;
                split2g(__uint_as_float(wr[j].x), __uint_as_float(wr[j].y), h01, l01);
                split2g(__uint_as_float(wr[j].z), __uint_as_float(wr[j].w), h23, l23);
                *reinterpret_cast<uint2*>(&Wh[wrow[j] * G2_LD + wk[j]]) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(&Wl[wrow[j] * G2_LD + wk[j]]) = make_uint2(l01, l23);
            }
        }
        __syncthreads();
        if (k0 + G2_BK < a.K) gload(k0 + G2_BK);
        const int r = lane & 31, kh = (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < G2_BK / 16; ks++) {
            FragG xh, xl;
            xh.q = *reinterpret_cast<const uint4*>(&Ah[(wave * 32 + r) * G2_LD + ks * 16 + kh]);
            xl.q = *reinterpret_cast<const uint4*>(&Al[(wave * 32 + r) * G2_LD + ks * 16 + kh]);
#pragma unroll
            for (int t = 0; t < NT; t++) {
                FragG wh;
                wh.q = *reinterpret_cast<const uint4*>(&Wh[(t * 32 + r) * G2_LD + ks * 16 + kh]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh.v, wh.v, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl.v, wh.v, acc[t], 0, 0, 0);
                if constexpr (!WBF16) {
                    FragG wl;
                    wl.q = *reinterpret_cast<const uint4*>(&Wl[(t * 32 + r) * G2_LD + ks * 16 + kh]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh.v, wl.v, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // epilogue: lane holds C[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31] of each 32x32 tile.
    // The 16 row addresses are computed once (one divide each) and reused by every column tile.
    int64_t rowoff[16];
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        int m = m0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        rowoff[reg] = m < a.M ? row_off(a.cmap, m) : -1;
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int n = n0 + t * 32 + (lane & 31);
        if (n >= a.N) continue;
        const float bias = a.bias ? a.bias[n] : 0.0f;
        const float addv = a.addvec ? a.addvec[n] : 0.0f;
        const float scl = a.scale ? a.scale[n] : 1.0f;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            if (rowoff[reg] < 0) continue;
            float v = acc[t][reg] + bias;
            const int64_t co = rowoff[reg] + n;
            switch (a.epi) {
                case EPI_NONE: break;
                case EPI_GELU: v = gelu1(v); break;
                case EPI_SILU: v = silu1(addv + v); break;
                case EPI_ELU: v = elu1(v); break;
                case EPI_RESADD: v = a.R[co] + v; break;
                case EPI_SCALE_RESADD: v = a.R[co] + scl * v; break;
                case EPI_GATE_RESADD: {
                    int m = m0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    v = a.R[co] + a.gate[(int64_t)m * a.ldg + n] * v;
                    break;
                }
                case EPI_AXPY: v = a.R[co] + a.alpha * v; break;
                case EPI_RESADD_ELU: v = elu1(a.R[co] + v); break;
            }
            a.C[co] = v;
        }
    }
}

bool gemm2_supported(const GemmArgs& a) {
    const int kalign = a.w_bf16 ? 8 : 4;
    return a.M >= 96 && a.K % kalign == 0 && aligned16(a.A) && a.amap.ld % 4 == 0 && a.amap.batch_stride % 4 == 0 &&
           a.ldw % kalign == 0 && aligned16(a.W);
}

template <int BN>
static void launch_bn(const GemmArgs& a, hipStream_t stream) {
    const int ncol = (a.N + BN - 1) / BN, npan = (a.M + G2_BM - 1) / G2_BM;
    dim3 grid((unsigned)(((npan + 7) / 8) * 8 * ncol));
    if (a.w_bf16) hipLaunchKernelGGL((k_gemm2<BN, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_gemm2<BN, false>), grid, dim3(256), 0, stream, a);
}

void launch_gemm2(const GemmArgs& a, hipStream_t stream) {
    note_launch("k_gemm2");
    if (a.N > 64) launch_bn<128>(a, stream);
    else if (a.N > 32) launch_bn<64>(a, stream);
    else launch_bn<32>(a, stream);
}

}  // namespace ptts
