// gemm_wres.hip -- many-row GEMM whose weights live in LDS for the lifetime of a block: the last two transposed convs of the
// SEANet decoder (mimi.go:740-788, convtranspose1d.go:73-148 as [rows x 2*Cin] x [2*Cin x stride*Cout] products, K = N = 256 and
// K = 512, N = 640) and the Mimi transformer's linear1 (K = 512, N = 2048, GELU; mimi.go:506-525).
#include <algorithm>
#include <cstdlib>

#include "kernels.h"
#include "device_util.h"

namespace ptts {

// C[M,N] = epi( A[M,K] * W[N,K]^T + bias ) for K <= 256, N <= 256, bf16 weights; numerics of k_gemm3 (activations split into
// bf16 hi + lo in registers, both multiply the bf16 weights, f32 accumulation, same k order).
//
// k_gemm3 walks such a product as 256-row tiles whose eight waves meet at a barrier for every 64..128 k of weights they stage
// through LDS; with K = 256 a tile is two such chunks between a prologue that waits for HBM and a 128-KB epilogue, and one
// block per CU (236 registers per lane) has nothing else to run meanwhile: 2.4 ms for 5.9 GB and 1 TFLOP at batch 64.
// Here the weights (N x K bf16 <= 128 KB) are copied into LDS ONCE per block, in the fragment layout of k_gemm3
// ([32-k group][column][32 k]), and after that single barrier the eight waves of a block never meet again: each walks its own
// 32-row panels (panel = (block, wave) + j * blocks * 8), keeps a ring of four 32-k steps of its rows in flight ACROSS panel
// boundaries (the rows are the only thing it ever fetches), multiplies against the resident fragments and stores its 32 x N
// results.  A wave that waits for memory leaves the matrix core to the other wave of its SIMD.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2w(float a, float b, unsigned& hi, unsigned& lo) {
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);
    f32x2 r = f - __builtin_convertvector(h, f32x2);
    bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

// chunk c (of the four 16-byte chunks of a column's 32 k) is stored at c ^ fw(column): with it the 16 lanes of each group a
// ds_read_b128 is served in hit 16 different bank slots (gemm5.hip has the derivation; the plain layout was a 2-way conflict)
__device__ __forceinline__ int wres_fw(int col) { return ((col >> 3) & 1) * 3; }

union FragW {
    bf16x8 v;
    uint4 q;
};

constexpr int WR_RING = 4, WR_NW = 8;

// A block owns the column tile [n0, n0 + WR_N) of the output, n0 = ct * WR_N, and a group of row panels; WR_N x WR_K bf16 = 128 KB.
// With more than one column tile (K = 512: 128-column tiles) the tiles of one row group are placed on ONE XCD
// (block = xcd + 8 * (ct + ntiles * local group)): they walk the same panels at the same pace, so the rows are fetched from HBM
// once and re-read from that XCD's L2 by the sibling tiles.
// Columns / k past the matrix are zero weights in LDS: their products vanish and are never stored.
template <int WR_N, int WR_K>
__global__ __launch_bounds__(WR_NW * 64) void k_gemm_wres(GemmArgs a, int ntiles, int rgl) {
    __shared__ __attribute__((aligned(16))) unsigned char Wl[WR_N * WR_K * 2];   // [k group (8)][column (256)][32 k bf16 = 64 B]
    __shared__ __attribute__((aligned(16))) float Bl[WR_N];                      // bias (a fetch from memory behind the row ring would wait for it)
    constexpr int NT = WR_N / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int ct = jb % ntiles, grp = xcd * rgl + jb / ntiles, ngrp = 8 * rgl;   // column tile, row group, row groups in all
    const int n0 = ct * WR_N;
    {   // ---- weights -> LDS, once ----
        constexpr int PPR = WR_K / 8;                 // 16-byte pieces per weight row
        for (int pc = tid; pc < WR_N * PPR; pc += WR_NW * 64) {
            const int n = pc / PPR, kk = pc % PPR;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (n0 + n < a.N && kk * 8 < a.K) u = *reinterpret_cast<const uint4*>((const char*)a.W + ((int64_t)(n0 + n) * a.ldw + kk * 8) * 2);
            *reinterpret_cast<uint4*>(Wl + (kk >> 2) * (WR_N * 64) + n * 64 + (((kk & 3) ^ wres_fw(n)) << 4)) = u;
        }
    }
    if (tid < WR_N) Bl[tid] = (a.bias && n0 + tid < a.N) ? a.bias[n0 + tid] : 0.0f;
    __syncthreads();

    const int npan = (a.M + 31) >> 5;
    const int stride = ngrp * WR_NW;                  // panels between two of this wave's
    const int p0 = grp * WR_NW + wave;
    if (p0 >= npan) return;                           // (after the only barrier)
    const int mine = (npan - p0 + stride - 1) / stride;
    const int nsteps = a.K >> 5;                      // K % 32 == 0 (host)
    const int total = mine * nsteps;

    // the wave's rows, one 32-k step at a time: lane = row r16 (two 16-row tiles), k group g -> two float4 (8 k).  The ring is
    // refilled in order, so the (panel, step) of the next request just advances: no division per request.
    float4 av[WR_RING][2][2];
    int lp = 0, li = 0;                               // panel ordinal and step of the next request
    const float* lrow[2];
    auto l_rows = [&]() {
        const int m0 = (p0 + lp * stride) << 5;
#pragma unroll
        for (int t = 0; t < 2; t++) lrow[t] = a.A + row_off(a.amap, min(m0 + t * 16 + r16, a.M - 1)) + g * 8;
    };
    l_rows();
    auto a_load = [&](float4 (&dst)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            dst[t][0] = *reinterpret_cast<const float4*>(lrow[t] + li * 32);
            dst[t][1] = *reinterpret_cast<const float4*>(lrow[t] + li * 32 + 4);
        }
        if (++li == nsteps) { li = 0; lp = min(lp + 1, mine - 1); l_rows(); }   // (past the last panel: its rows again, never used)
    };
#pragma unroll
    for (int r = 0; r < WR_RING; r++) a_load(av[r]);

    f32x4 acc[2][NT];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int n = 0; n < NT; n++) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    int step = 0, pj = 0;                             // step within the panel, panel ordinal
    for (int q0 = 0; q0 < total; q0 += WR_RING) {
#pragma unroll
        for (int r = 0; r < WR_RING; r++) {
            const int q = q0 + r;
            if (q < total) {                          // wave-uniform
                FragW ah[2], al[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const float4 x0 = av[r][t][0], x1 = av[r][t][1];
                    split2w(x0.x, x0.y, ah[t].q.x, al[t].q.x);
                    split2w(x0.z, x0.w, ah[t].q.y, al[t].q.y);
                    split2w(x1.x, x1.y, ah[t].q.z, al[t].q.z);
                    split2w(x1.z, x1.w, ah[t].q.w, al[t].q.w);
                }
                a_load(av[r]);                            // unconditional (no load behind a branch: the compiler would drain the ring there)
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* wb = Wl + step * (WR_N * 64) + r16 * 64 + ((g ^ wres_fw(r16)) << 4);
                FragW wh[2];                              // fragment n + 1 is requested before fragment n is multiplied
                wh[0].q = *reinterpret_cast<const uint4*>(wb);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    if (n + 1 < NT) wh[(n + 1) & 1].q = *reinterpret_cast<const uint4*>(wb + (n + 1) * 1024);
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n & 1].v, ah[t].v, acc[t][n], 0, 0, 0);
                        PTTS_LO_MFMA(acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n & 1].v, al[t].v, acc[t][n], 0, 0, 0));
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (++step == nsteps) {               // the panel is complete: bias, epilogue, store, start over
                    const int m0 = (p0 + pj * stride) << 5;
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const int m = m0 + t * 16 + r16;
                        const int64_t ro = row_off(a.cmap, min(m, a.M - 1));
#pragma unroll
                        for (int n = 0; n < NT; n++) {
                            const int lc = n * 16 + 4 * g, col = n0 + lc;
                            float4 v = make_float4(acc[t][n][0], acc[t][n][1], acc[t][n][2], acc[t][n][3]);
                            acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (m >= a.M || col >= a.N) continue;   // N % 4 == 0 (host)
                            const float4 b = *reinterpret_cast<const float4*>(Bl + lc);
                            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                            if (a.epi == EPI_ELU) { v.x = elu_fast(v.x); v.y = elu_fast(v.y); v.z = elu_fast(v.z); v.w = elu_fast(v.w); }
                            else if (a.epi == EPI_GELU) { v.x = gelu1(v.x); v.y = gelu1(v.y); v.z = gelu1(v.z); v.w = gelu1(v.w); }
                            *reinterpret_cast<float4*>(a.C + ro + col) = v;
                        }
                    }
                    step = 0;
                    pj++;
                }
            }
        }
    }
}

static bool wres_common(const GemmArgs& a) {
    return a.w_bf16 && a.K % 32 == 0 && a.N % 4 == 0 && a.M >= 2048 && a.aop == AOP_NONE && (a.epi == EPI_NONE || a.epi == EPI_ELU || a.epi == EPI_GELU) &&
           !a.rope_cos && !a.kslice && !a.tail && aligned16(a.A) && a.amap.ld % 4 == 0 && a.amap.batch_stride % 4 == 0 && a.ldw % 8 == 0 &&
           aligned16(a.W) && aligned16(a.C) && a.cmap.ld % 4 == 0 && a.cmap.batch_stride % 4 == 0;
}
static int wres_shape(const GemmArgs& a) {   // 1: 256 x 256 tile, 2: 128 columns x 512 k, 0: not this kernel's
    if (!wres_common(a)) return 0;
    if (a.K <= 256 && a.K > 128 && a.N <= 256 && a.N > 128) return 1;              // (narrower shapes would multiply zero padding)
    if (a.K <= 512 && a.K > 256 && a.N >= 512 && (a.N + 127) / 128 <= 32 && a.M >= 16384) return 2;
    return 0;
}
bool gemm_wres_supported(const GemmArgs& a) { return wres_shape(a) != 0; }

void launch_gemm_wres(const GemmArgs& a, hipStream_t stream) {
    static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const int per_xcd = std::max(1, cus / 8);
    note_launch(wres_shape(a) == 1 ? "k_gemm_wres<256,256>" : "k_gemm_wres<128,512>");
    if (wres_shape(a) == 1) {
        const int npan = (a.M + 31) / 32;
        const int rgl = std::max(1, std::min(per_xcd, (npan + 8 * WR_NW - 1) / (8 * WR_NW)));   // one block per CU at most
        hipLaunchKernelGGL((k_gemm_wres<256, 256>), dim3((unsigned)(8 * rgl)), dim3(WR_NW * 64), 0, stream, a, 1, rgl);
    } else {
        const int ntiles = (a.N + 127) / 128;
        const int rgl = std::max(1, per_xcd / ntiles);                                            // row groups per XCD
        hipLaunchKernelGGL((k_gemm_wres<128, 512>), dim3((unsigned)(8 * rgl * ntiles)), dim3(WR_NW * 64), 0, stream, a, ntiles, rgl);
    }
}

}  // namespace ptts
