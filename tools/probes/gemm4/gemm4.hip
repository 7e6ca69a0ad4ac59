// gemm4.hip -- the many-row GEMM of the Mimi decoder for its deep, wide shapes (K >= 256, N a multiple of 256, bf16
// weights): the decoder transformer's qkv / out_proj / linear1 / linear2, the first SEANet convolution and the first
// transposed convolution (mimi.go:719-789, conv1d.go:20-83, convtranspose1d.go:73-148) at M = 10^5 rows.
//
// Same numerics as k_gemm3 (f32 activations split into bf16 hi + lo in registers, both multiplied by the bf16 weights on
// v_mfma_f32_16x16x32_bf16, f32 accumulation, the same k order -> equal bits), different data movement:
//   * BOTH operands go through LDS, filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) in
//     full 128-byte lines: an activation piece is 8 rows x 128 B (32 k of f32), a weight piece 16 columns x 64 B.  k_gemm3
//     loads the activations fragment-shaped (16 rows x 64 B per instruction) straight into registers, which keeps the
//     texture-address unit twice as busy per byte and leaves the loads' latency to a two-step register ring;
//   * the LDS image is XOR-swizzled so that every ds_read_b128 of a fragment is conflict-free; with LDS-DMA the
//     destination is lane-linear, so the swizzle sits on the SOURCE address of each lane and on the read address;
//   * a block is 4 waves x 32 rows = 128 rows x 256 columns with two 32-k stages (64 KB): two blocks per CU, so one
//     block's prologue / epilogue / barrier waits run under the other's MFMAs.
#include "kernels.h"
#include "device_util.h"
#include <algorithm>

namespace ptts {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ void split2g(float a, float b, unsigned& hi, unsigned& lo) {
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);
    f32x2 r = f - __builtin_convertvector(h, f32x2);
    bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

// XOR swizzles of the LDS images, chosen for the lane groups a ds_read_b128 is served in on gfx950 ({0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} and the same + 32: every group holds each of the 16 fragment rows once, rows 4..11 with the
// neighbouring k group): with them the 16 lanes of a group hit 16 different 16-byte bank slots.
__device__ __forceinline__ int g4_hbit(int r) { return ((r >> 2) ^ (r >> 3)) & 1; }                         // 1 for fragment rows 4..11
__device__ __forceinline__ int g4_fa(int row) { return ((row >> 1) & 7) ^ (g4_hbit(row & 15) << 1); }       // activation row: chunk c (of 8) at c ^ fa
__device__ __forceinline__ int g4_fw(int col) { return ((col >> 3) & 1) * 3; }                              // weight column: chunk c (of 4) at c ^ fw

union Frag4 {
    bf16x8 v;
    uint4 q;
};

constexpr int G4_BM = 128, G4_BN = 256, G4_BK = 32, G4_NW = 4;
constexpr int G4_ASTAGE = G4_BM * G4_BK * 4;   // 16 KB: [row][128 B], 16-byte chunk c of row r stored at c ^ g4_fa(r)
constexpr int G4_WSTAGE = G4_BN * G4_BK * 2;   // 16 KB: [column][64 B], chunk c of column n stored at c ^ g4_fw(n)
constexpr int G4_STAGE = G4_ASTAGE + G4_WSTAGE;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const void* g, char* lds_piece) {
    // One wave instruction: lane l's 16 bytes land at lds_piece + 16 l (the destination is M0 = a wave-uniform LDS address, + lane x 16).
    // Inline asm, not __builtin_amdgcn_global_load_lds: the compiler treats the builtin as an LDS store that may alias every later
    // ds_read and puts s_waitcnt vmcnt(0) in front of the first one, which serialises DMA and MFMA.  The kernel orders the DMA
    // itself: a counted wait before the barrier that precedes the first read of a stage.
    const unsigned dst = (unsigned)(uintptr_t)lds_piece;   // LDS addresses are the low 32 bits of the generic pointer
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
}

// bias / RoPE / residual forms / store of four consecutive output columns of row m (the same code as k_gemm3's epilogue)
__device__ __forceinline__ void g4_store(const GemmArgs& a, int m, int64_t ro, int col, float4 v) {
    if (a.bias) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (a.rope_cos && col < a.rope_cols) {   // the lane's four columns are two (even, odd) pairs of one head
        const int pos = a.rope_row_pos ? a.rope_row_pos[m] : a.rope_pos0 + (a.rope_rows_per_seg ? m % a.rope_rows_per_seg : m);
        const int half = a.rope_hd >> 1, j = (col % a.rope_hd) >> 1;
        const float2 cs = *reinterpret_cast<const float2*>(a.rope_cos + (int64_t)pos * half + j);
        const float2 sn = *reinterpret_cast<const float2*>(a.rope_sin + (int64_t)pos * half + j);
        const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
        v.x = x0 * cs.x - x1 * sn.x; v.y = x0 * sn.x + x1 * cs.x;
        v.z = x2 * cs.y - x3 * sn.y; v.w = x2 * sn.y + x3 * cs.y;
    }
    const int64_t co = ro + col;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.epi >= EPI_RESADD) r = *reinterpret_cast<const float4*>(a.R + co);
    switch (a.epi) {
        case EPI_NONE: break;
        case EPI_GELU: v.x = gelu1(v.x); v.y = gelu1(v.y); v.z = gelu1(v.z); v.w = gelu1(v.w); break;
        case EPI_SILU: {
            float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.addvec) ad = *reinterpret_cast<const float4*>(a.addvec + col);
            v.x = silu1(ad.x + v.x); v.y = silu1(ad.y + v.y); v.z = silu1(ad.z + v.z); v.w = silu1(ad.w + v.w);
            break;
        }
        case EPI_ELU: v.x = elu_fast(v.x); v.y = elu_fast(v.y); v.z = elu_fast(v.z); v.w = elu_fast(v.w); break;
        case EPI_RESADD: v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w; break;
        case EPI_SCALE_RESADD: {
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
            if (a.scale) sc = *reinterpret_cast<const float4*>(a.scale + col);
            v.x = r.x + sc.x * v.x; v.y = r.y + sc.y * v.y; v.z = r.z + sc.z * v.z; v.w = r.w + sc.w * v.w;
            break;
        }
        case EPI_GATE_RESADD: {
            const float4 gt = *reinterpret_cast<const float4*>(a.gate + (int64_t)m * a.ldg + col);
            v.x = r.x + gt.x * v.x; v.y = r.y + gt.y * v.y; v.z = r.z + gt.z * v.z; v.w = r.w + gt.w * v.w;
            break;
        }
        case EPI_RESADD_ELU: v.x = elu_fast(r.x + v.x); v.y = elu_fast(r.y + v.y); v.z = elu_fast(r.z + v.z); v.w = elu_fast(r.w + v.w); break;
        case EPI_AXPY: v.x = r.x + a.alpha * v.x; v.y = r.y + a.alpha * v.y; v.z = r.z + a.alpha * v.z; v.w = r.w + a.alpha * v.w; break;
    }
    *reinterpret_cast<float4*>(a.C + co) = v;
}

}  // namespace

template <bool ELU>
__global__ __launch_bounds__(G4_NW * 64, 2) void k_gemm4(GemmArgs a) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * G4_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    // XCD-aware order (speed only): the column tiles of a 128-row panel run back to back on one XCD, whose L2 then holds the panel
    const int ncol = a.N / G4_BN, npan = (a.M + G4_BM - 1) / G4_BM;
    const int bid = blockIdx.x, xcd = bid & 7, jb = bid >> 3;
    const int pan = (jb / ncol) * 8 + xcd;
    if (pan >= npan) return;
    const int m0 = pan * G4_BM, n0 = (jb % ncol) * G4_BN;

    // ---- LDS-DMA sources of this lane: four activation pieces and four weight pieces per stage and wave -------------------
    const char* asrc[4];
    const char* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);                 // row of the block this lane fetches for
        const int c = (lane & 7) ^ g4_fa(row);                             // ... and the 16-byte chunk of its 128-byte k slab
        asrc[i] = reinterpret_cast<const char*>(a.A + row_off(a.amap, min(m0 + row, a.M - 1))) + c * 16;
        const int col = (wave * 4 + i) * 16 + (lane >> 2);
        const int cw = (lane & 3) ^ g4_fw(col);
        wsrc[i] = reinterpret_cast<const char*>(a.W) + ((int64_t)(n0 + col) * a.ldw) * 2 + cw * 16;
    }
    auto stage_load = [&](int s, int stage) {
        char* ab = lds + stage * G4_STAGE + wave * 4096;
        char* wb = lds + stage * G4_STAGE + G4_ASTAGE + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; i++) glds16(asrc[i] + (int64_t)s * 128, ab + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; i++) glds16(wsrc[i] + (int64_t)s * 64, wb + i * 1024);
    };

    f32x4 acc[2][16];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int n = 0; n < 16; n++) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read addresses inside a stage (constant per lane)
    const int fa = g4_fa(r16);
    int a_off[2][2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int row = wave * 32 + t * 16 + r16;
        a_off[t][0] = row * 128 + (((2 * g) ^ fa) << 4);
        a_off[t][1] = row * 128 + (((2 * g + 1) ^ fa) << 4);
    }
    const int w_off = G4_ASTAGE + r16 * 64 + ((g ^ g4_fw(r16)) << 4);

    const int nsteps = a.K >> 5;
    stage_load(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < nsteps; s++) {
        if (s + 1 < nsteps) stage_load(s + 1, (s + 1) & 1);   // (issued between the MFMAs instead: slower here -- the last pieces land too late)
        const char* st = lds + (s & 1) * G4_STAGE;
        Frag4 ah[2], al[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            float4 x0 = *reinterpret_cast<const float4*>(st + a_off[t][0]);
            float4 x1 = *reinterpret_cast<const float4*>(st + a_off[t][1]);
            if constexpr (ELU) {
                x0.x = elu_fast(x0.x); x0.y = elu_fast(x0.y); x0.z = elu_fast(x0.z); x0.w = elu_fast(x0.w);
                x1.x = elu_fast(x1.x); x1.y = elu_fast(x1.y); x1.z = elu_fast(x1.z); x1.w = elu_fast(x1.w);
            }
            split2g(x0.x, x0.y, ah[t].q.x, al[t].q.x);
            split2g(x0.z, x0.w, ah[t].q.y, al[t].q.y);
            split2g(x1.x, x1.y, ah[t].q.z, al[t].q.z);
            split2g(x1.z, x1.w, ah[t].q.w, al[t].q.w);
        }
#pragma unroll
        for (int n = 0; n < 16; n++) {
            Frag4 wh;
            wh.q = *reinterpret_cast<const uint4*>(st + w_off + n * 1024);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, ah[t].v, acc[t][n], 0, 0, 0);
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, al[t].v, acc[t][n], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next stage have landed
        __syncthreads();                                   // ... so have everyone's, and everyone is done reading this stage
    }

    // epilogue: lane holds C[row r16 of tile t][columns n*16 + 4g .. +3]
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int m = m0 + wave * 32 + t * 16 + r16;
        if (m >= a.M) continue;
        const int64_t ro = row_off(a.cmap, m);
#pragma unroll
        for (int n = 0; n < 16; n++) {
            const int col = n0 + n * 16 + 4 * g;
            g4_store(a, m, ro, col, make_float4(acc[t][n][0], acc[t][n][1], acc[t][n][2], acc[t][n][3]));
        }
    }
}

// ---- persistent form: 8 waves x 32 rows = 256 rows x 256 columns, three 32-k stages (144 KB), one block per CU walking its
// tiles; the DMA of the step two ahead is issued at every step -- across tile boundaries too, so a tile's first stages land
// under the previous tile's last MFMAs and its epilogue -- and retired by a COUNTED wait (the six pieces of the following step
// stay in flight across the barrier).
constexpr int G4P_BM = 256, G4P_NW = 8, G4P_ASTAGE = G4P_BM * G4_BK * 4 /* 32 KB */, G4P_STAGE = G4P_ASTAGE + G4_WSTAGE /* 48 KB */;

__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    // as glds16, with the address as a wave-uniform base (SGPR pair) + a per-lane 32-bit byte offset
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

struct G4Src {
    unsigned a[4], w[2];   // byte offsets of this lane's sources from a.A / a.W at k = 0
};

template <bool ELU, bool STAMP>
__global__ __launch_bounds__(G4P_NW * 64, 2) void k_gemm4p(GemmArgs a) {
    // stamps (tools/stamps_gemm4.py): shader-clock ticks of every wave of block 0 at the phase boundaries of each step of its first two tiles
    int g4_ti = 0;
#define G4_STAMP(j, s) do { if (STAMP && blockIdx.x == 0 && lane == 0 && g4_ti < 2 && (s) < 64) a.dbg[(((g4_ti * 64 + (s)) * 8 + wave) * 8) + (j)] = (j) == 7 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
    __shared__ __attribute__((aligned(1024))) char lds[3 * G4P_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int ncol = a.N / G4_BN, npan = (a.M + G4P_BM - 1) / G4P_BM;
    const int U = ((npan + 7) / 8) * 8 * ncol, G = gridDim.x;
    // tile u -> (row panel, column tile), XCD-aware as in k_gemm3: u & 7 is constant for a block (G % 8 == 0), and the column
    // tiles of a panel are walked by neighbouring blocks of one XCD at the same time
    auto tile_of = [&](int u, int& m0, int& n0) {
        const int jb = u >> 3, pan = (jb / ncol) * 8 + (u & 7);
        m0 = pan * G4P_BM; n0 = (jb % ncol) * G4_BN;
        return pan < npan;
    };
    auto next_tile = [&](int u, int& m0, int& n0) {
        do { u += G; } while (u < U && !tile_of(u, m0, n0));
        return u;
    };
    auto make_src = [&](int m0, int n0, G4Src& o) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = (wave * 4 + i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ g4_fa(row);
            o.a[i] = (unsigned)(row_off(a.amap, min(m0 + row, a.M - 1)) * 4 + c * 16);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int col = (wave * 2 + i) * 16 + (lane >> 2);
            const int cw = (lane & 3) ^ g4_fw(col);
            o.w[i] = (unsigned)(((int64_t)(n0 + col) * a.ldw) * 2 + cw * 16);
        }
    };
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    auto stage_load = [&](const G4Src& src, int s, int stage) {
        const char* ab = reinterpret_cast<const char*>(a.A) + (int64_t)s * 128;
        const char* wb = reinterpret_cast<const char*>(a.W) + (int64_t)s * 64;
        const unsigned la = lds0 + stage * G4P_STAGE + wave * 4096, lw = lds0 + stage * G4P_STAGE + G4P_ASTAGE + wave * 2048;
#pragma unroll
        for (int i = 0; i < 4; i++) glds16s(ab, src.a[i], la + i * 1024);
#pragma unroll
        for (int i = 0; i < 2; i++) glds16s(wb, src.w[i], lw + i * 1024);
    };

    int m0, n0, u = blockIdx.x;
    if (!tile_of(u, m0, n0)) u = next_tile(u, m0, n0);
    if (u >= U) return;
    G4Src cur, nxt;
    make_src(m0, n0, cur);

    const int fa = g4_fa(r16);
    int a_off[2][2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int row = wave * 32 + t * 16 + r16;
        a_off[t][0] = row * 128 + (((2 * g) ^ fa) << 4);
        a_off[t][1] = row * 128 + (((2 * g + 1) ^ fa) << 4);
    }
    const int w_off = G4P_ASTAGE + r16 * 64 + ((g ^ g4_fw(r16)) << 4);

    const int nsteps = a.K >> 5;   // >= 8
    stage_load(cur, 0, 0);
    stage_load(cur, 1, 1);
    int stg = 0;   // stage of the step about to be multiplied
    for (;;) {
        int m1, n1;
        const int un = next_tile(u, m1, n1);
        const bool has_next = un < U;
        if (has_next) make_src(m1, n1, nxt);
        f32x4 acc[2][16];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int n = 0; n < 16; n++) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nsteps; s++) {
            // this step's pieces have landed once all but the six youngest DMAs (the following step's) are done
            G4_STAMP(0, s); G4_STAMP(7, s);
            // (lgkmcnt(0): every LDS read of the previous step has returned before this wave lets the others refill that stage)
            if (s + 1 < nsteps || has_next) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            G4_STAMP(1, s);
            __builtin_amdgcn_s_barrier();   // ... everyone's have, and everyone has finished reading the stage refilled next
            asm volatile("" ::: "memory");
            G4_STAMP(2, s);
            // the DMA of the step two ahead (six pieces per wave) is issued BETWEEN this step's MFMAs, one piece every other
            // column tile: an LDS-DMA costs its wave ~100 cycles of issue, which the other wave of the SIMD fills with MFMAs
            const int sf = stg == 0 ? 2 : stg - 1;   // (stg + 2) % 3
            const bool in_tile = s + 2 < nsteps, do_load = in_tile || has_next;
            const int s2 = in_tile ? s + 2 : s + 2 - nsteps;
            unsigned po[6];
#pragma unroll
            for (int i = 0; i < 4; i++) po[i] = in_tile ? cur.a[i] : nxt.a[i];
#pragma unroll
            for (int i = 0; i < 2; i++) po[4 + i] = in_tile ? cur.w[i] : nxt.w[i];
            const char* pab = reinterpret_cast<const char*>(a.A) + (int64_t)s2 * 128;
            const char* pwb = reinterpret_cast<const char*>(a.W) + (int64_t)s2 * 64;
            const unsigned pla = lds0 + sf * G4P_STAGE + wave * 4096, plw = lds0 + sf * G4P_STAGE + G4P_ASTAGE + wave * 2048;
            G4_STAMP(3, s);
            const char* st = lds + stg * G4P_STAGE;
            Frag4 ah[2], al[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                float4 x0 = *reinterpret_cast<const float4*>(st + a_off[t][0]);
                float4 x1 = *reinterpret_cast<const float4*>(st + a_off[t][1]);
                if constexpr (ELU) {
                    x0.x = elu_fast(x0.x); x0.y = elu_fast(x0.y); x0.z = elu_fast(x0.z); x0.w = elu_fast(x0.w);
                    x1.x = elu_fast(x1.x); x1.y = elu_fast(x1.y); x1.z = elu_fast(x1.z); x1.w = elu_fast(x1.w);
                }
                split2g(x0.x, x0.y, ah[t].q.x, al[t].q.x);
                split2g(x0.z, x0.w, ah[t].q.y, al[t].q.y);
                split2g(x1.x, x1.y, ah[t].q.z, al[t].q.z);
                split2g(x1.z, x1.w, ah[t].q.w, al[t].q.w);
            }
#pragma unroll
            for (int n = 0; n < 16; n++) {
                Frag4 wh;
                wh.q = *reinterpret_cast<const uint4*>(st + w_off + n * 1024);
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, ah[t].v, acc[t][n], 0, 0, 0);
                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, al[t].v, acc[t][n], 0, 0, 0);
                }
                if ((n & 1) && n < 12 && do_load) {
                    const int i = n >> 1;
                    if (i < 4) glds16s(pab, po[i], pla + i * 1024);
                    else glds16s(pwb, po[i], plw + (i - 4) * 1024);
                }
            }
            stg = stg == 2 ? 0 : stg + 1;
            G4_STAMP(4, s);
        }
        G4_STAMP(5, 63);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int m = m0 + wave * 32 + t * 16 + r16;
            if (m >= a.M) continue;
            const int64_t ro = row_off(a.cmap, m);
#pragma unroll
            for (int n = 0; n < 16; n++) g4_store(a, m, ro, n0 + n * 16 + 4 * g, make_float4(acc[t][n][0], acc[t][n][1], acc[t][n][2], acc[t][n][3]));
        }
        G4_STAMP(6, 63);
        g4_ti++;
        if (!has_next) break;
        u = un; m0 = m1; n0 = n1; cur = nxt;
    }
}

bool gemm4_supported(const GemmArgs& a) {
    const bool res = a.epi >= EPI_RESADD;
    return a.w_bf16 && !a.kslice && a.M >= 16384 && a.K >= 256 && a.K % 32 == 0 && a.N % G4_BN == 0 && aligned16(a.A) && a.amap.ld % 4 == 0 &&
           a.amap.batch_stride % 4 == 0 && a.ldw % 8 == 0 && aligned16(a.W) && aligned16(a.C) && a.cmap.ld % 4 == 0 && a.cmap.batch_stride % 4 == 0 &&
           (!a.bias || aligned16(a.bias)) && (!a.addvec || aligned16(a.addvec)) && (!a.scale || aligned16(a.scale)) && (!res || aligned16(a.R)) &&
           (a.epi != EPI_GATE_RESADD || (aligned16(a.gate) && a.ldg % 4 == 0)) &&
           (!a.rope_cos || (a.rope_hd % 4 == 0 && a.rope_cols % 4 == 0 && a.epi == EPI_NONE));
}

thread_local int g_gemm4_cfg = 0;   // debug knob (ptts_debug_gemm): 1 = the two-stage, two-blocks-per-CU form

static bool gemm4p_offsets_fit(const GemmArgs& a) {   // the persistent form addresses A and W with 32-bit byte offsets
    const int64_t last = a.amap.rows_per_batch ? ((int64_t)(a.M - 1) / a.amap.rows_per_batch) * a.amap.batch_stride + ((int64_t)(a.M - 1) % a.amap.rows_per_batch) * a.amap.ld
                                               : (int64_t)(a.M - 1) * a.amap.ld;
    return (last + a.K) * 4 < ((int64_t)1 << 32) && (int64_t)a.N * a.ldw * 2 < ((int64_t)1 << 32);
}

void launch_gemm4(const GemmArgs& a, hipStream_t stream) {
    note_launch(a.rope_cos ? "k_gemm4+rope" : "k_gemm4");
    if (g_gemm4_cfg != 1 && gemm4p_offsets_fit(a)) {
        static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n / 8 * 8; }();
        const int ncol = a.N / G4_BN, npan = (a.M + G4P_BM - 1) / G4P_BM, U = ((npan + 7) / 8) * 8 * ncol;
        dim3 grid((unsigned)std::min(U, cus));
        if (a.dbg) hipLaunchKernelGGL((k_gemm4p<false, true>), grid, dim3(G4P_NW * 64), 0, stream, a);
        else if (a.aop == AOP_ELU) hipLaunchKernelGGL((k_gemm4p<true, false>), grid, dim3(G4P_NW * 64), 0, stream, a);
        else hipLaunchKernelGGL((k_gemm4p<false, false>), grid, dim3(G4P_NW * 64), 0, stream, a);
        return;
    }
    const int ncol = a.N / G4_BN, npan = (a.M + G4_BM - 1) / G4_BM;
    dim3 grid((unsigned)(((npan + 7) / 8) * 8 * ncol));
    static const int pad = [] { const char* e = getenv("PTTS_G4_PAD"); return e ? atoi(e) : 0; }();   // diagnosis: extra LDS per block (40000: one block per CU)
    if (a.aop == AOP_ELU) hipLaunchKernelGGL(k_gemm4<true>, grid, dim3(G4_NW * 64), pad, stream, a);
    else hipLaunchKernelGGL(k_gemm4<false>, grid, dim3(G4_NW * 64), pad, stream, a);
}

}  // namespace ptts
