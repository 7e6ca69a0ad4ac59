// tall.hip -- the AR step's big linears at 128..256 rows: a row-preparation kernel and a 64 x 64-tile GEMM whose operands both stream through LDS by LDS-DMA.
#include <hip/hip_ext.h>

#include "../../include/ptts.h"
#include "common.h"
#include "kernels.h"
#include "device_util.h"

namespace ptts {

// ------------------------------------------------------------------------------------------------
// Why a second kernel family for the same linears (flow_transformer.go:326-389: LN -> in_proj, LN -> linear1 -> GELU -> linear2).
// k_skinny (skinny.hip) is built for the regime where a launch is ONE block per CU: 16 rows x 64 columns per block, the LayerNorm / split-K prologue redone by
// every column block, the 128-KB weight tile re-read (from L2) by every 16-row tile.  Its time is 4.4 us + (bytes its CUs take in) / 70 GB/s per CU, and past
// 64 rows the bytes grow with the row tiles: in_proj 8.5 -> 14.1 -> 21.0 us at 64 / 128 / 256 rows (profiles/r5_wide_by_grid_b*.txt).  At 128+ rows the
// shape is a GEMM, so it is tiled like one:
//   * k_rowprep: the prologue ONCE per row instead of once per (row, column block): x' = x + sum of the pending split-K planes (+ bias), written back; LayerNorm
//     (biased variance, linear.go:295-309; k_skinny's arithmetic, so a row normalises to the same bits either way); the result split into bf16 hi + lo PLANES
//     [rows][K] (x = hi + lo to ~2^-17) -- what k_skinny's prologue leaves in its LDS image, in global memory instead.
//   * k_tall: a block = 64 rows x 64 columns x a K slice; its 16 waves are 4 row tiles x 4 column tiles, each wave owns one 16 x 16 output tile over the whole slice.
//     Per 128-deep K chunk the block needs 32 KB of row planes and 16 KB of weights (the fragment-ordered copy of model.cpp add_tiled: 1-KB pieces that ARE the
//     MFMA operand of a wave): 48 pieces of 1 KB, three per wave, copied global -> LDS by LDS-DMA into a three-stage ring, one barrier per chunk.  The row image
//     is XOR-swizzled on the SOURCE side of the copy (a DMA lane may read any 16 bytes), so that the fragment reads are conflict-free (swizzle found with the bank
//     model of tools/probes/lds_conflicts.py: k_skinny's own image has a 2-way conflict there).  Per 64 x 64 outputs the block takes in 384 KB (K = 1024) against
//     4 x 256 KB for the four k_skinny blocks it replaces.
// Arithmetic per product is k_skinny's: two v_mfma_f32_16x16x32_bf16 per 32-deep step (hi x w, lo x w, separate accumulators, added at the end), the same k
// permutation inside a 128-deep super-step (lane group q owns k = 32 q + 8 s + j in MFMA s); operands swapped so that a lane ends up with four consecutive
// columns of one row (16-byte stores).  What differs from the 64-row path is the grouping of the sums over K (one accumulator pair over the whole slice instead of
// four K parts): rounding order, covered by the same tolerances (tests/test_gpu_wide_batch.py runs every 64-row check at 128 and 256 rows through these kernels).
// ------------------------------------------------------------------------------------------------
typedef __bf16 tl_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tl_bf16x2 __attribute__((ext_vector_type(2)));
typedef float tl_f32x2 __attribute__((ext_vector_type(2)));
typedef float tl_f32x4 __attribute__((ext_vector_type(4)));

constexpr int TL_COLS = 64, TL_KC = 128;                     // a block's columns and the K chunk; its rows are the template parameter ROWS (64, or 32 up to 128 rows:
                                                             // twice the blocks, each taking in two thirds of the bytes -- a 128-row launch has 96..128 blocks of 64 rows)
constexpr int TL_W = TL_COLS * TL_KC * 2;                    // bytes of the weight pieces of a stage (16 KB)
constexpr int TL_LDS = 160 * 1024;
template <int ROWS> struct TlShape {
    static constexpr int PLANE = ROWS * TL_KC * 2;           // bytes of one row plane of a stage (16 KB at 64 rows)
    static constexpr int STAGE = 2 * PLANE + TL_W;           // 48 KB / 32 KB
    static constexpr int NS = TL_LDS / STAGE;                // ring depth: 3 / 5 stages
    static constexpr int WAVES = ROWS / 16 * 4;              // 16 / 8
    static constexpr int APIECES = 2 * ROWS / 4 / WAVES;     // 1-KB pieces of the two row planes per wave and chunk (2)
    static constexpr int WPIECES = 16 / WAVES;               // ... of the weights (1 / 2)
    static constexpr int PIECES = APIECES + WPIECES;
};

union TlFrag {
    tl_bf16x8 v;
    uint4 q;
};

__device__ __forceinline__ void tl_split2(float a, float b, unsigned& hi, unsigned& lo) {   // (skinny.hip split2)
    tl_f32x2 f = {a, b};
    tl_bf16x2 h = __builtin_convertvector(f, tl_bf16x2);
    tl_f32x2 r = f - __builtin_convertvector(h, tl_f32x2);
    tl_bf16x2 l = __builtin_convertvector(r, tl_bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

// 16-byte chunk swizzle of a row image (row r of a 16-row tile, 16 chunks of 8 k per 128-deep chunk): chunk c sits at c ^ tl_swz(r).  With a 256-byte row pitch the
// 16 lanes one ds_read_b128 service group holds (two lane groups q of different rows) then hit 16 different 16-byte slots of the 256-byte bank period.
__device__ __forceinline__ int tl_swz(int r) { return ((r >> 2) & 1) | (r & 2) | ((r & 1) << 3); }

// One LDS-DMA wave-instruction: lane l's 16 bytes at src_lane -> lds_base + 16 l (ffn_fused.hip ff_dma1k: inline assembly so that the compiler's wait counters do
// not serialise the ring; the wait state between the write of m0 and its use is written out; m0 is reserved to the compiler, which keeps nothing in it across statements)
__device__ __forceinline__ void tl_dma1k(const char* src_lane, char* dst) {
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(base), "v"(src_lane) : "memory");
}

template <bool PLANES, int ROWS>
__global__ __launch_bounds__(ROWS * 16) void k_tall(TallArgs a) {
    using S = TlShape<ROWS>;
    constexpr int TL_PLANE = S::PLANE, TL_STAGE = S::STAGE, TL_NS = S::NS, TL_ROWS = ROWS;
    __shared__ __attribute__((aligned(16))) char lds[TL_NS * TL_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, rt = wave >> 2;                 // this wave's column tile and row tile inside the block
    const int i16 = lane & 15, q = lane >> 4;
    const int m_blk = blockIdx.y * TL_ROWS, n_blk = blockIdx.x * TL_COLS, z = blockIdx.z;
    const int nss_all = (a.K + 127) >> 7;
    const int ss_per = a.splitk > 1 ? (nss_all + a.splitk - 1) / a.splitk : nss_all;
    const int ss0 = z * ss_per, nch = max(0, min(nss_all, ss0 + ss_per) - ss0);

    // ---- the three pieces this wave copies per chunk ----
    // row planes, piece `wave`: rows 4 wave .. 4 wave + 3 of the block (256 B each in the image); lane -> (row, 16-byte slot); the slot holds source chunk slot ^ swz(row)
    const int prow = 4 * wave + (lane >> 4);
    const int64_t grow = min(m_blk + prow, a.M - 1);          // (rows past M repeat the last row: computed, never stored)
    const int pchunk = (lane & 15) ^ tl_swz(prow & 15);
    const char* a_hi = reinterpret_cast<const char*>(a.ah) + (grow * a.lda + pchunk * 8) * 2;
    const char* a_lo = reinterpret_cast<const char*>(a.al) + (grow * a.lda + pchunk * 8) * 2;
    // weights: 16 pieces per chunk (column tile p >> 2 of the block, MFMA step p & 3 of the chunk's super-step; model.cpp add_tiled: [tile][ss][4][64 lanes] x 16 B);
    // wave w copies piece w (and w + 8 when the block has 8 waves)
    const char* w_src[S::WPIECES];
#pragma unroll
    for (int i = 0; i < S::WPIECES; i++) {
        const int p = wave + i * S::WAVES;
        const int wtile = min(blockIdx.x * 4 + (p >> 2), (a.N + 15) / 16 - 1);
        w_src[i] = reinterpret_cast<const char*>(a.Wt) + ((((int64_t)wtile * nss_all) * 4 + (p & 3)) * 64 + lane) * 16;
    }
    auto issue = [&](int ch) {
        char* st = lds + (ch % TL_NS) * TL_STAGE;
        const int ss = ss0 + ch;
        tl_dma1k(a_hi + (int64_t)ss * (TL_KC * 2), st + wave * 1024);
        tl_dma1k(a_lo + (int64_t)ss * (TL_KC * 2), st + TL_PLANE + wave * 1024);
#pragma unroll
        for (int i = 0; i < S::WPIECES; i++) tl_dma1k(w_src[i] + (int64_t)ss * 4096, st + 2 * TL_PLANE + (wave + i * S::WAVES) * 1024);
    };

    // ---- epilogue operands: requested before the copies (they are older than every copy in the wave's queue, so the counted waits below cover them) ----
    const int m = m_blk + rt * 16 + i16, n = n_blk + cg * 16 + q * 4;
    const bool m_ok = m < a.M, n_ok = n < a.N;
    const int64_t mc = m_ok ? m : 0;
    const int nc = n_ok ? n : 0;
    float4 e_bias = make_float4(0.f, 0.f, 0.f, 0.f), e_r = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool epi_ops = a.splitk <= 1 || z == 0;
    if (epi_ops) {
        if (a.bias) e_bias = *reinterpret_cast<const float4*>(a.bias + nc);
        if (a.R) e_r = *reinterpret_cast<const float4*>(a.R + mc * a.ldr + nc);
    }

#pragma unroll
    for (int c = 0; c < TL_NS - 1; c++) if (c < nch) issue(c);
    tl_f32x4 acc_h = {0.f, 0.f, 0.f, 0.f}, acc_l = {0.f, 0.f, 0.f, 0.f};
    const int x_off = (rt * 16 + i16) * 256, x_swz = tl_swz(i16);
    for (int ch = 0; ch < nch; ch++) {
        // chunk ch has landed: this wave's copies of it (everything but the copies of the chunks issued after it -- up to NS - 2 of them -- is complete), then everybody's
        switch (min(TL_NS - 2, nch - 1 - ch) * S::PIECES) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        }
        __syncthreads();   // ... which also says that every wave has finished reading chunk ch - 1: its stage takes chunk ch + NS - 1
        if (ch + TL_NS - 1 < nch) issue(ch + TL_NS - 1);
        const char* st = lds + (ch % TL_NS) * TL_STAGE;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            TlFrag w, xh, xl;
            w.q = *reinterpret_cast<const uint4*>(st + 2 * TL_PLANE + (cg * 4 + s) * 1024 + lane * 16);
            const int off = x_off + (((q * 4 + s) ^ x_swz) << 4);
            xh.q = *reinterpret_cast<const uint4*>(st + off);
            xl.q = *reinterpret_cast<const uint4*>(st + TL_PLANE + off);
            acc_h = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, xh.v, acc_h, 0, 0, 0);
            acc_l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, xl.v, acc_l, 0, 0, 0);
        }
    }
    // D of the swapped product: lane (i16, q) holds row m = its x row, columns n .. n + 3
    const tl_f32x4 acc = acc_h + acc_l;
    if (!m_ok || !n_ok) return;
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (a.splitk > 1) {   // raw sums of the K slice; slice 0 carries residual + bias (k_skinny's convention: the consumer reads plane 0 as its rows and adds the others)
        if (z == 0 && a.R) { v[0] = e_r.x + (v[0] + e_bias.x); v[1] = e_r.y + (v[1] + e_bias.y); v[2] = e_r.z + (v[2] + e_bias.z); v[3] = e_r.w + (v[3] + e_bias.w); }
        *reinterpret_cast<float4*>(a.partial + (int64_t)z * a.zstride + (int64_t)m * a.N + n) = make_float4(v[0], v[1], v[2], v[3]);
        return;
    }
    v[0] += e_bias.x; v[1] += e_bias.y; v[2] += e_bias.z; v[3] += e_bias.w;
    switch (a.epi) {
        case EPI_NONE: break;
        case EPI_GELU:
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = gelu1(v[e]);
            break;
        case EPI_RESADD:
            v[0] = e_r.x + v[0]; v[1] = e_r.y + v[1]; v[2] = e_r.z + v[2]; v[3] = e_r.w + v[3];
            break;
    }
    if constexpr (PLANES) {
        unsigned h01, l01, h23, l23;
        tl_split2(v[0], v[1], h01, l01);
        tl_split2(v[2], v[3], h23, l23);
        *reinterpret_cast<uint2*>(a.ch + (int64_t)m * a.ldp + n) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(a.cl + (int64_t)m * a.ldp + n) = make_uint2(l01, l23);
    } else {
        *reinterpret_cast<float4*>(a.C + (int64_t)m * a.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

bool tall_supported(const TallArgs& a) {
    if (!a.ah || !a.al || !a.Wt || a.M <= 0 || a.M > kStepMaxRows || a.N <= 0 || a.K <= 0) return false;
    if (a.K % TL_KC || a.N % 4 || a.lda % 8 || !aligned16(a.ah) || !aligned16(a.al) || !aligned16(a.Wt)) return false;
    if (a.bias && !aligned16(a.bias)) return false;
    if (a.R && (a.ldr % 4 || !aligned16(a.R))) return false;
    if (a.splitk > 1) return a.partial && aligned16(a.partial) && a.zstride % 4 == 0 && a.epi == EPI_NONE && !a.ch;
    if (a.epi != EPI_NONE && a.epi != EPI_GELU && a.epi != EPI_RESADD) return false;
    if (a.epi == EPI_RESADD && !a.R) return false;
    if (a.ch) return a.cl && a.ldp % 4 == 0 && (reinterpret_cast<uintptr_t>(a.ch) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.cl) & 7) == 0;
    return a.C && a.ldc % 4 == 0 && aligned16(a.C);
}

void launch_tall(const TallArgs& a, hipStream_t stream) {
    note_launch("k_tall");
    const int rows = a.M <= 128 ? 32 : 64;
    const dim3 grid((unsigned)((a.N + TL_COLS - 1) / TL_COLS), (unsigned)((a.M + rows - 1) / rows), (unsigned)std::max(1, a.splitk));
    if (rows == 32) {
        if (a.ch) hipLaunchKernelGGL((k_tall<true, 32>), grid, dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((k_tall<false, 32>), grid, dim3(512), 0, stream, a);
    } else {
        if (a.ch) hipLaunchKernelGGL((k_tall<true, 64>), grid, dim3(1024), 0, stream, a);
        else hipLaunchKernelGGL((k_tall<false, 64>), grid, dim3(1024), 0, stream, a);
    }
}

// ------------------------------------------------------------------------------------------------
// k_rowprep: one wave per row.  x' = x + (sum of `psplit` planes, in order) + pbias -> x_out; LayerNorm(x') (affine) -> bf16 hi / lo planes (+ the f32 rows to y_out).
// The arithmetic is k_skinny's fused prologue (PRO_LN | PRO_AFFINE | PRO_PARTIAL), instruction for instruction: a row normalises to the same bits on either path.
// ------------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(256) void k_rowprep(PrepArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wave;
    if (m >= a.M) return;
    float4 xr[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) xr[j] = *reinterpret_cast<const float4*>(a.x + (int64_t)m * a.ldx + (lane + 64 * j) * 4);
    float4 lw[NJ], lb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) { lw[j] = *reinterpret_cast<const float4*>(a.ln_w + (lane + 64 * j) * 4); lb[j] = *reinterpret_cast<const float4*>(a.ln_b + (lane + 64 * j) * 4); }
    if (a.partial) {
        float4 ps[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) ps[j] = *reinterpret_cast<const float4*>(a.partial + (int64_t)m * a.D + (lane + 64 * j) * 4);
        for (int zz = 1; zz < a.psplit; zz++) {
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const float4 p = *reinterpret_cast<const float4*>(a.partial + (int64_t)zz * a.pstride + (int64_t)m * a.D + (lane + 64 * j) * 4);
                ps[j].x += p.x; ps[j].y += p.y; ps[j].z += p.z; ps[j].w += p.w;
            }
        }
        if (a.pbias) {
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const float4 p = *reinterpret_cast<const float4*>(a.pbias + (lane + 64 * j) * 4);
                ps[j].x += p.x; ps[j].y += p.y; ps[j].z += p.z; ps[j].w += p.w;
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) { xr[j].x += ps[j].x; xr[j].y += ps[j].y; xr[j].z += ps[j].z; xr[j].w += ps[j].w; }
        if (a.x_out) {
#pragma unroll
            for (int j = 0; j < NJ; j++) *reinterpret_cast<float4*>(a.x_out + (int64_t)m * a.D + (lane + 64 * j) * 4) = xr[j];
        }
    }
    tl_f32x2 s2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; j++) s2 += tl_f32x2{xr[j].x, xr[j].z} + tl_f32x2{xr[j].y, xr[j].w};
    const float rk = __builtin_amdgcn_rcpf((float)a.D);
    const float mean = wave_sum_dpp(s2.x + s2.y) * rk;
    const tl_f32x2 m2 = {mean, mean};
    tl_f32x2 v2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const tl_f32x2 da = tl_f32x2{xr[j].x, xr[j].y} - m2, db = tl_f32x2{xr[j].z, xr[j].w} - m2;
        v2 += da * da + db * db;
    }
    const float inv_std = __builtin_amdgcn_rsqf(wave_sum_dpp(v2.x + v2.y) * rk + a.eps);
    const tl_f32x2 is2 = {inv_std, inv_std};
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        tl_f32x2 oa = (tl_f32x2{xr[j].x, xr[j].y} - m2) * is2, ob = (tl_f32x2{xr[j].z, xr[j].w} - m2) * is2;
        oa = oa * tl_f32x2{lw[j].x, lw[j].y} + tl_f32x2{lb[j].x, lb[j].y};
        ob = ob * tl_f32x2{lw[j].z, lw[j].w} + tl_f32x2{lb[j].z, lb[j].w};
        const int col = (lane + 64 * j) * 4;
        if (a.y_out) *reinterpret_cast<float4*>(a.y_out + (int64_t)m * a.D + col) = make_float4(oa.x, oa.y, ob.x, ob.y);
        unsigned h01, l01, h23, l23;
        tl_split2(oa.x, oa.y, h01, l01);
        tl_split2(ob.x, ob.y, h23, l23);
        *reinterpret_cast<uint2*>(a.yh + (int64_t)m * a.ldy + col) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(a.yl + (int64_t)m * a.ldy + col) = make_uint2(l01, l23);
    }
}

bool rowprep_supported(const PrepArgs& a) {
    return a.x && a.ln_w && a.ln_b && a.yh && a.yl && a.M > 0 && (a.D == 512 || a.D == 1024) && a.ldx % 4 == 0 && a.ldy % 4 == 0 && aligned16(a.x) && aligned16(a.ln_w) &&
           aligned16(a.ln_b) && (reinterpret_cast<uintptr_t>(a.yh) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.yl) & 7) == 0 &&
           (!a.partial || (a.psplit >= 1 && aligned16(a.partial) && a.pstride % 4 == 0)) && (!a.pbias || aligned16(a.pbias)) && (!a.x_out || aligned16(a.x_out)) &&
           (!a.y_out || aligned16(a.y_out));
}

void launch_rowprep(const PrepArgs& a, hipStream_t stream) {
    note_launch("k_rowprep");
    const dim3 grid((unsigned)((a.M + 3) / 4));
    if (a.D == 1024) hipLaunchKernelGGL(k_rowprep<4>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(k_rowprep<2>, grid, dim3(256), 0, stream, a);
}

}  // namespace ptts
