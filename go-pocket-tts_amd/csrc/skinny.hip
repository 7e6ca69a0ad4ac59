// skinny.hip -- the AR step's weight-streaming linear (the kernel the roofline in bench.py prices).
#include <hip/hip_ext.h>

#include "../../include/ptts.h"
#include "common.h"
#include "kernels.h"
#include "device_util.h"
#include "step_open.h"

namespace ptts {

// ------------------------------------------------------------------------------------------------
// Weight-streaming linear for the AR step (K2-K4, K8, K9, K11 at M = batch <= kStepMaxRows rows: one 16-row tile per blockIdx.y).
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

constexpr int SK_CHAIN_BX = 3;  // blocks per row tile of a chained last launch: one for fx and the books, two for the halves of x (1024 columns)
constexpr int SK_KMAX = 1024;   // K slice per block (LDS image: 2 x 16 rows x 2 KB = 64 KB)
constexpr int SK_KMAX2 = 2048;  // ... for the plain (no fused prologue) split-K launches: NJ = 8, image 2 x 16 rows x 4 KB = 128 KB

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

// prologue variants (template parameter PRO)
constexpr int PRO_LN = 1, PRO_AFFINE = 2, PRO_MOD = 4, PRO_PARTIAL = 8, PRO_ONE = 16;   // PRO_ONE (with PRO_PARTIAL): exactly one plane (the usual case: see below)

// NJ: 256-column groups of the K slice per lane (2: K slice <= 512, 4: <= 1024, 8: <= 2048 -- halves the split-K planes the
// consumer of a 4096-deep product has to re-read).
// CG: 16-column groups per block (4: 64 columns x 4 K parts; 1: 16 columns x 16 K parts -- four times the blocks, for
// launches whose 64-column grid would leave most of the chip idle: batch <= 16, the reference's own batch-1 case)
// FIN: the launch carries the AR step's bookkeeping (SkinnyFuse::fin) -- a template parameter so that the other variants do
// not hold its eight registers (the fused-prologue variants sit at the 128-register limit of a 16-wave block)
// WT: how the weights are stored -- 0: f32, 1: bf16, 2: per-row-scaled int8 (offset-binary bytes; converted to bf16 in registers,
// which is exact for [-127, 127]; the row scale is applied to the sums in the epilogue)
// CHAIN (with FIN): the block also opens the NEXT step -- see SkinnyFuse::chain and step_open.h
template <int WT, bool STAMP, int PRO, int NJ, int CG, bool FIN = false, bool CHAIN = false>
__global__ __launch_bounds__(1024) void k_skinny(const void* p_wt, const float* p_a, int p_lda, int p_ms, int p_n, int p_k, const float* p_part, const float* p_lnw,
                                                 const float* p_lnb, GemmArgs a, SkinnyFuse fu, float* partial, unsigned long long* stamps) {
    // The nine leading scalars -- copies of a.Wt, a.A, a.amap.ld, a.M | split << 16, a.N, a.K, fu.partial, fu.ln_w, fu.ln_b: 14 dwords, all
    // the user SGPRs a kernel can have preloaded (-amdgpu-kernarg-preload-count) -- arrive in registers with the dispatch, so every
    // request on the critical path of the launch leaves before the first scalar-cache round trip for the argument block has
    // returned (that block is cold on every dispatch: ~0.5-1 us).
    // ORDER OF THE REQUESTS.  A wave's loads return in issue order (vmcnt counts down oldest first): whatever is requested after the
    // weights cannot be consumed before the last weight byte of the wave has landed, and the weights are the long pole (32-128 KB per
    // block from HBM: 2-5 us).  So the tile's rows, the first split-K plane and the LayerNorm vectors go out FIRST, the weights second:
    // the whole prologue (sum, LayerNorm, hi/lo split, LDS image, the block's barrier) then runs while the weights stream, and each
    // wave starts multiplying when its own weights arrive.  (Round 2 had the weights first: in-situ stamps showed every wave's
    // prologue starting only when its weights were in, profiles/r3_step_stamps_weights_first.txt.)
    const int p_m = p_ms & 0xffff, splitk = p_ms >> 16;
    // stamps (tools/microbench.py only; null in the product): shader-clock ticks of wave 0 of every block at the phase boundaries
#define SK_STAMP(i) do { if (STAMP && threadIdx.x == 0) stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = (i) == 7 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
    SK_STAMP(0);
    SK_STAMP(7);
    constexpr int WV = WT == 2 ? 2 : (WT == 1 ? 4 : 8);   // 16-byte weight loads per lane per super-step
    // LDS image: row r (0..15) = 2048 B = 128 chunks of 16 B; chunk c is stored at c ^ r, so the 16 lanes that read
    // the same logical chunk of 16 different rows hit 16 different bank groups
    constexpr int RB = NJ > 4 ? 4096 : 2048, CMASK = RB / 16 - 1;   // bytes and 16-byte chunks (- 1) per image row
    __shared__ __attribute__((aligned(16))) unsigned char Xh[16 * RB];
    __shared__ __attribute__((aligned(16))) unsigned char Xl[16 * RB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar branches
    constexpr int KP = 16 / CG;                // K parts per column group
    const int cg = wave % CG, kq4 = wave / CG;
    // (a chained launch runs SK_CHAIN_BX blocks per row tile that all compute the one column block of the frame -- same instructions, same bits -- and
    // share out the next step's opening: block 0 keeps the books, stores the frame and computes fx, the others the columns of x)
    const int cbx = CHAIN ? 0 : blockIdx.x;
    const int n = cbx * (16 * CG) + cg * 16 + (lane & 15);
    const int m0 = blockIdx.y * 16, z = blockIdx.z, q = lane >> 4;
    const bool n_ok = n < p_n;
    const int kper = splitk > 1 ? ((p_k + splitk - 1) / splitk + 127) / 128 * 128 : p_k;
    const int k_begin = z * kper, k_end = min(p_k, k_begin + kper);
    const int klen = k_end - k_begin;          // <= SK_KMAX (host guarantees)
    const int nss = (klen + 127) >> 7;         // 128-deep super-steps in the slice (<= 8)
    const int ssq = (nss + KP - 1) / KP;       // super-steps per K part (<= NTW)
    const int ss_lo = kq4 * ssq;

    // ---- activations: wave w stages row w of the tile; lane owns float4 columns lane + 64 j.  Requested first (see above): their
    // addresses come from preloaded scalars alone ----
    // Loads are unconditional within the row (a column >= klen replays column 0) and the value is masked afterwards.
    // A wave whose row does not exist (batch not a multiple of 16: batch 1 has 15 of them) skips the prologue altogether: its
    // LDS row feeds only output rows that are never stored, and the SIMD it shares is left to the waves with real rows.
    const bool row_ok = m0 + wave < p_m;   // wave-uniform
    const int64_t mrow = row_ok ? m0 + wave : 0;
    float4 xr[NJ], ps[(PRO & PRO_PARTIAL) ? NJ : 1];
    int kc[NJ];
    bool kok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int k = (lane + 64 * j) * 4;
        kok[j] = k < klen;
        kc[j] = kok[j] ? k : 0;
    }
    if (row_ok) {
#pragma unroll
        for (int j = 0; j < NJ; j++) xr[j] = *reinterpret_cast<const float4*>(p_a + mrow * p_lda + k_begin + kc[j]);
        if constexpr ((PRO & PRO_PARTIAL) != 0) {   // the first plane of the pending split-K sum (dense [psplit][M][K]: K is the row width here)
#pragma unroll
            for (int j = 0; j < NJ; j++) ps[j] = *reinterpret_cast<const float4*>(p_part + mrow * p_k + kc[j]);
        }
    }
    // LayerNorm weight / bias of the lane's columns, with the rows (sharing one copy per block through LDS was measured: the extra
    // barrier in front of the prologue cost more than the 120 KB of L1 hits it saved, profiles/r3_step_ab.txt)
    float4 lwr[(PRO & PRO_AFFINE) ? NJ : 1], lbr[(PRO & PRO_AFFINE) ? NJ : 1];
    if constexpr ((PRO & PRO_AFFINE) != 0) {
        if (row_ok) {
#pragma unroll
            for (int j = 0; j < NJ; j++) { lwr[j] = *reinterpret_cast<const float4*>(p_lnw + kc[j]); lbr[j] = *reinterpret_cast<const float4*>(p_lnb + kc[j]); }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // The CU's vector-memory port serves its waves' requests in the order they were issued, across waves: without this barrier
    // wave 15's rows queue behind the weights of waves 0..14 (120 KB), and the prologue of the block is as late as its last wave.
    // (a tile of a few rows -- the reference's own batch-1 case -- has nothing worth waiting for: its weights go out at once)
    if (p_m - m0 > 4) __builtin_amdgcn_s_barrier();
    // ---- weights: everything this wave will multiply is requested now ----
    // Fragment-ordered copy (model.cpp add_tiled, zero-padded to 16 columns x 128 k): one contiguous 1-KiB burst per
    // wave-instruction.  The loads are unconditional (a tile / super-step that does not exist re-reads block 0 and is never
    // multiplied or stored): a load under a condition becomes a branch, and the compiler then drains the memory pipe
    // (s_waitcnt vmcnt(0)) between the weight loads.
    constexpr int NTW = (2 * NJ + KP - 1) / KP;   // super-steps per K part: the slice has at most 2 NJ of them
    uint4 w[NTW][WV];
    {
        const int nss_all = (p_k + 127) >> 7;
        const int tile = cbx * CG + cg, ss_base = k_begin >> 7;
        const bool tile_ok = tile * 16 < p_n;
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const int ss = ss_lo + t;
            const bool ok = tile_ok && t < ssq && ss < nss;
            const uint4* src = reinterpret_cast<const uint4*>(p_wt) + (ok ? (((int64_t)tile * nss_all + ss_base + ss) * WV) * 64 : 0) + lane;
#pragma unroll
            for (int s = 0; s < WV; s++) w[t][s] = src[s * 64];
        }
    }
    // chained opening of the next step: this wave's weight fragments (and bias pieces) of the two 32-deep linears go out right behind
    // the step weights -- they depend on nothing the step computes, and the four blocks of this launch have nobody to hide a late
    // request behind (a first cut that requested them after the products lost 5 us per step: three dependent cold round trips at the tail)
    constexpr int CT = 2;               // 16-column tiles per wave and pass: block 0 covers 16 x 2 x 16 = 512 columns of fx, blocks 1.. 512 columns of x each
    constexpr bool CWB = WT != 0;       // the 32-deep linears are bf16 whenever the step weights are not f32 (host: StepChain::w_bf16)
    [[maybe_unused]] SoW<CWB> cw[CHAIN ? CT : 1];
    [[maybe_unused]] float2 c_vs = make_float2(0.f, 0.f);
    [[maybe_unused]] const bool c_fx = blockIdx.x == 0;                           // this block's share: fx, or a slice of x
    [[maybe_unused]] const int c_t0 = c_fx ? 0 : ((int)blockIdx.x - 1) * 16 * CT;   // its first 16-column tile
    // what the tail needs of the chain record, held from here on: a second look at fu.fin->ch behind the block's barriers would be a fresh (cold) scalar
    // round trip on the tail of a launch that has nothing to hide it behind -- and so would the bos vector
    [[maybe_unused]] float* c_out = nullptr; [[maybe_unused]] float* c_x0 = nullptr;
    [[maybe_unused]] const void* c_wp = nullptr; [[maybe_unused]] const float* c_bp = nullptr;
    [[maybe_unused]] int c_n = 0;
    [[maybe_unused]] float c_bos = 0.f;
    if constexpr (CHAIN) {
        const StepChain& ch = fu.fin->ch;
        c_wp = c_fx ? ch.w_pj : ch.w_in;
        c_bp = c_fx ? ch.b_pj : ch.b_in;
        c_n = c_fx ? ch.d_pj : ch.d_in;
        c_out = c_fx ? ch.fx : ch.x;
        c_x0 = ch.x0;
#pragma unroll
        for (int t = 0; t < CT; t++) cw[t] = so_load_w<CWB>(c_wp, c_bp, min((c_t0 + wave * CT + t) * 16, c_n - 16), c_n, lane);
        c_bos = ch.bos[min(n, SO_K - 1)];
    }
    __builtin_amdgcn_sched_barrier(0);   // rows, then weights, then everything that needs the argument block
    SK_STAMP(1);
    // ---- prologue: the rest of what the rows need (LayerNorm vectors, further planes) and the arithmetic; the prologue variant is a
    // template parameter so that a launch only holds the vectors it uses (16 waves per CU leave 128 VGPRs per lane) ----
    if (row_ok) {
        const int m = m0 + wave;
        const bool m_ok = true;
        // Every other kernel argument is pulled into SGPRs here, in one batch of scalar loads.  The argument block of a fresh
        // dispatch is cold in the scalar cache and each miss is a round trip to memory (~0.5 us); left to itself the compiler
        // loads a field where it is first used, behind a branch on an earlier field, which chained five to six such round
        // trips through the kernel (measured with tools/stamps_skinny.py: 2.3 us before the first weight load was issued).
        // Only what the variant reads is pinned: everything listed is live in SGPRs at this point at once, and the most-launched
        // fused-prologue variant was spilling 15 scalars to vector lanes (v_writelane / v_readlane on the critical path) to make room
        // for fields it never touches (the copies of Wt / A / lda / M / N / K arrive as leading arguments).
        asm volatile("" ::"s"(a.bias), "s"(a.addvec), "s"(a.C), "s"(a.cmap.ld), "s"(a.R), "s"(a.scale), "s"(a.gate), "s"(a.ldg), "s"(a.alpha),
                     "s"(a.tail), "s"(a.epi), "s"(partial));
        if constexpr ((PRO & PRO_PARTIAL) != 0) asm volatile("" ::"s"(fu.psplit), "s"(fu.pstride), "s"(fu.pbias), "s"(fu.x_out));
        if constexpr ((PRO & PRO_LN) != 0) asm volatile("" ::"s"(fu.eps), "s"(fu.y_out));
        if constexpr ((PRO & PRO_MOD) != 0) asm volatile("" ::"s"(fu.shift), "s"(fu.scale), "s"(fu.ldmod));
        if constexpr (FIN) asm volatile("" ::"s"(fu.fin));
        float4 lc[(PRO & PRO_MOD) ? NJ : 1], lh[(PRO & PRO_MOD) ? NJ : 1];
        auto load_params = [&]() {
            if constexpr ((PRO & PRO_MOD) != 0) {
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    lc[j] = *reinterpret_cast<const float4*>(fu.scale + mrow * fu.ldmod + kc[j]);
                    lh[j] = *reinterpret_cast<const float4*>(fu.shift + mrow * fu.ldmod + kc[j]);
                }
            }
        };
        if constexpr ((PRO & PRO_PARTIAL) != 0) {   // rows += sum_z partial[z] (+ bias); the planes are added in a fixed order
            // (the AR step hands over [x + (sums_0 + bias)] as the rows and the sums of the other K slices as planes: runtime.cpp step_core)
            const float* pp = p_part + mrow * p_k;
            if constexpr ((PRO & PRO_ONE) == 0) {
                for (int z0 = 1; z0 < fu.psplit; z0 += 3) {   // three more slices per round trip
                    float4 p[3][NJ];
#pragma unroll
                    for (int u = 0; u < 3; u++) {
                        const int zz = min(z0 + u, fu.psplit - 1);
#pragma unroll
                        for (int j = 0; j < NJ; j++) p[u][j] = *reinterpret_cast<const float4*>(pp + (int64_t)zz * fu.pstride + kc[j]);
                    }
#pragma unroll
                    for (int u = 0; u < 3; u++) {
                        if (z0 + u < fu.psplit) {
#pragma unroll
                            for (int j = 0; j < NJ; j++) { ps[j].x += p[u][j].x; ps[j].y += p[u][j].y; ps[j].z += p[u][j].z; ps[j].w += p[u][j].w; }
                        }
                    }
                }
            }
            load_params();   // after the partials: the register file (128 per lane at 16 waves) does not hold both sets in flight
            if (fu.pbias) {
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    const float4 p = *reinterpret_cast<const float4*>(fu.pbias + kc[j]);
                    ps[j].x += p.x; ps[j].y += p.y; ps[j].z += p.z; ps[j].w += p.w;
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                if (m_ok && kok[j]) { xr[j].x += ps[j].x; xr[j].y += ps[j].y; xr[j].z += ps[j].z; xr[j].w += ps[j].w; }
                else xr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (fu.x_out && blockIdx.x == 0 && m_ok) {
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    if (kok[j]) *reinterpret_cast<float4*>(fu.x_out + (int64_t)m * p_k + (lane + 64 * j) * 4) = xr[j];
            }
        } else {
            load_params();
            __builtin_amdgcn_sched_barrier(0);   // every request of the prologue is in flight before the first value is consumed
#pragma unroll
            for (int j = 0; j < NJ; j++)
                if (!(m_ok && kok[j])) xr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr ((PRO & PRO_LN) != 0) {   // LayerNorm over the full row (K == row width), biased variance (linear.go:295-309)
            // two elements per instruction where the ISA has it (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): the prologue runs on
            // four waves per SIMD and its vector instructions are on the critical path of the launch
            f32x2 s2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; j++) s2 += f32x2{xr[j].x, xr[j].z} + f32x2{xr[j].y, xr[j].w};
            // 1/K by v_rcp_f32 (exact for the power-of-two widths of the model, 1 ulp otherwise) and 1/sqrt by v_rsq_f32 (1 ulp)
            // instead of two IEEE divisions and a square root: ~35 vector instructions less on the critical path of every row
            const float rk = __builtin_amdgcn_rcpf((float)p_k);
            const float mean = wave_sum_dpp(s2.x + s2.y) * rk;
            const f32x2 m2 = {mean, mean};
            f32x2 v2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                if (kok[j]) {
                    const f32x2 da = f32x2{xr[j].x, xr[j].y} - m2, db = f32x2{xr[j].z, xr[j].w} - m2;
                    v2 += da * da + db * db;
                }
            }
            const float inv_std = __builtin_amdgcn_rsqf(wave_sum_dpp(v2.x + v2.y) * rk + fu.eps);
            const f32x2 is2 = {inv_std, inv_std};
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                f32x2 oa = (f32x2{xr[j].x, xr[j].y} - m2) * is2, ob = (f32x2{xr[j].z, xr[j].w} - m2) * is2;
                if constexpr ((PRO & PRO_AFFINE) != 0) {
                    const float4 lw = lwr[j], lb = lbr[j];
                    oa = oa * f32x2{lw.x, lw.y} + f32x2{lb.x, lb.y};
                    ob = ob * f32x2{lw.z, lw.w} + f32x2{lb.z, lb.w};
                }
                if constexpr ((PRO & PRO_MOD) != 0) {
                    const f32x2 one = {1.0f, 1.0f};
                    oa = oa * (f32x2{lc[j].x, lc[j].y} + one) + f32x2{lh[j].x, lh[j].y};
                    ob = ob * (f32x2{lc[j].z, lc[j].w} + one) + f32x2{lh[j].z, lh[j].w};
                }
                float4 o = make_float4(oa.x, oa.y, ob.x, ob.y);
                if (!kok[j]) o = make_float4(0.f, 0.f, 0.f, 0.f);
                xr[j] = o;
                if (fu.y_out && blockIdx.x == 0 && m_ok && kok[j]) *reinterpret_cast<float4*>(fu.y_out + (int64_t)m * p_k + (lane + 64 * j) * 4) = o;
            }
        }
        SK_STAMP(2);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int k = (lane + 64 * j) * 4;
            if (k >= nss * 128) continue;
            unsigned h01, l01, h23, l23;
            split2(xr[j].x, xr[j].y, h01, l01);
            split2(xr[j].z, xr[j].w, h23, l23);
            const int off = wave * RB + ((((k >> 3) ^ wave) & CMASK) << 4) + ((k & 4) << 1);
            *reinterpret_cast<uint2*>(&Xh[off]) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(&Xl[off]) = make_uint2(l01, l23);
        }
    }
    __syncthreads();
    SK_STAMP(3);
    // ---- epilogue operands: requested by the waves that will store (K quarter 0) now that the image is staged -- behind the weights in
    // the wave's queue, so that the prologue above never waits for anything younger than the rows (these loads sit under branches:
    // the compiler's wait counts across them are conservative) -- and consumed after the K reduction ----
    // (an element of R that aliases C is read and written by the same lane only)
    float e_bias = 0.f, e_addv = 0.f, e_scl = 1.f, e_r[4] = {0.f, 0.f, 0.f, 0.f}, e_g[4] = {0.f, 0.f, 0.f, 0.f};
    float e_ws = 1.0f;   // int8 weights: the scale of this lane's output column
    if constexpr (WT == 2) { if (kq4 == 0) e_ws = a.wscale[n_ok ? n : 0]; }
    // split launches (raw sums per K slice): slice 0 carries residual + bias when a.R is set, so that the consumer of the planes
    // reads [R + (sums_0 + bias)] + sums_1 ... -- one row image fewer than residual, planes and bias separately
    const bool epi_ops = kq4 == 0 && (splitk <= 1 || (z == 0 && a.R != nullptr));
    if (epi_ops) {
        const int nc = n_ok ? n : 0;
        if (a.bias) e_bias = a.bias[nc];
        if (a.addvec) e_addv = a.addvec[(a.tail && nc == p_n - 1) ? 0 : nc];
        if (a.scale) e_scl = a.scale[nc];
        if (a.epi >= EPI_RESADD || splitk > 1) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) e_r[reg] = a.R[(int64_t)min(m0 + q * 4 + reg, p_m - 1) * a.cmap.ld + nc];
        }
        if (a.epi == EPI_GATE_RESADD) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) e_g[reg] = a.gate[(int64_t)min(m0 + q * 4 + reg, p_m - 1) * a.ldg + nc];
        }
    }
    // fused step bookkeeping: every storing wave reads its rows' counters now; the single lane per row that advances them does
    // so after the block's last barrier, by which time these reads have returned (forced below)
    int f_act[FIN ? 4 : 1] = {0}, f_step[FIN ? 4 : 1] = {0};
    int f_cd[FIN ? 4 : 1] = {0}, f_fae[FIN ? 4 : 1] = {0}, f_max[FIN ? 4 : 1] = {0}, f_kv[FIN ? 4 : 1] = {0};   // the bookkeeping lane's operands,
    float f_logit[FIN ? 4 : 1] = {0.f}, f_thr[FIN ? 4 : 1] = {0.f};                                             // requested now, used at the very end
    if constexpr (FIN) {
        if (kq4 == 0 && splitk <= 1) {
            const StepState& fs = fu.fin->s;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int mm = min(m0 + q * 4 + reg, p_m - 1);
                f_act[reg] = fs.active[mm];
                f_step[reg] = fs.step[mm];
                f_cd[reg] = fs.countdown[mm];
                f_fae[reg] = fs.frames_after_eos[mm];
                f_max[reg] = fs.max_steps[mm];
                f_kv[reg] = fs.kv_len[mm];
                f_logit[reg] = fu.fin->eos_logit[mm];
                f_thr[reg] = fs.eos_threshold[mm];
            }
        }
    }

    // chained opening of the next step: waves 4..11 stage the next noise row (thread -> row, column pair); the row's counters are read
    // here, before the block's next barrier, like the bookkeeping lanes' -- the lane that advances them writes after the last one
    int c_st = 0, c_max = 0, c_act = 0;
    [[maybe_unused]] const int c_e = tid - 256, c_r = (c_e >> 4) & 15, c_c = (c_e & 15) * 2;
    [[maybe_unused]] const bool c_noise = CHAIN && blockIdx.x == 0 && tid >= 256 && tid < 512;
    if constexpr (CHAIN) {
        if (c_noise && fu.chain_noise && m0 + c_r < p_m) {
            const StepState& fs = fu.fin->s;
            c_st = fs.step[m0 + c_r];
            c_max = fs.max_steps[m0 + c_r];
            c_act = fs.active[m0 + c_r];
        }
    }

    f32x4 acc_h = {0.f, 0.f, 0.f, 0.f}, acc_l = {0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15;
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        if (t >= ssq || ss_lo + t >= nss) break;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int c = (ss_lo + t) * 16 + q * 4 + s;            // logical 16-byte chunk = 8 k
            const int off = i16 * RB + (((c ^ i16) & CMASK) << 4);
            Frag8 xh, xl;
            xh.q = *reinterpret_cast<const uint4*>(&Xh[off]);
            xl.q = *reinterpret_cast<const uint4*>(&Xl[off]);
            if constexpr (WT == 2) {
                // 8 offset-binary bytes -> 8 bf16: (float)byte - 128 is an integer in [-127, 127], exact in bf16
                const uint4 r = w[t][s >> 1];
                const unsigned d0 = (s & 1) ? r.z : r.x, d1 = (s & 1) ? r.w : r.y;
                Frag8 wv;
                f32x2 p;
                p = f32x2{(float)(d0 & 0xffu), (float)((d0 >> 8) & 0xffu)} - f32x2{128.f, 128.f};
                { bf16x2 b = __builtin_convertvector(p, bf16x2); wv.u[0] = *reinterpret_cast<unsigned*>(&b); }
                p = f32x2{(float)((d0 >> 16) & 0xffu), (float)(d0 >> 24)} - f32x2{128.f, 128.f};
                { bf16x2 b = __builtin_convertvector(p, bf16x2); wv.u[1] = *reinterpret_cast<unsigned*>(&b); }
                p = f32x2{(float)(d1 & 0xffu), (float)((d1 >> 8) & 0xffu)} - f32x2{128.f, 128.f};
                { bf16x2 b = __builtin_convertvector(p, bf16x2); wv.u[2] = *reinterpret_cast<unsigned*>(&b); }
                p = f32x2{(float)((d1 >> 16) & 0xffu), (float)(d1 >> 24)} - f32x2{128.f, 128.f};
                { bf16x2 b = __builtin_convertvector(p, bf16x2); wv.u[3] = *reinterpret_cast<unsigned*>(&b); }
                acc_h = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh.v, wv.v, acc_h, 0, 0, 0);
                acc_l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl.v, wv.v, acc_l, 0, 0, 0);
            } else if constexpr (WT == 1) {
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
    // ---- sum the K parts of each column group (fixed order) through a region of its own: no wave has to wait for the others to
    // have finished reading the activation image before it parks its sums ----
    if (STAMP) { if (acc_h[0] == 1.2345e-30f) SK_STAMP(6); }   // (stamp build) make stamp 4 wait for the MFMA results
    SK_STAMP(4);
    __shared__ float4 red[(KP - 1) * CG * 64];   // [kq4 - 1][cg][lane]
    const f32x4 accv = acc_h + acc_l;
    if (kq4 > 0) red[((kq4 - 1) * CG + cg) * 64 + lane] = make_float4(accv[0], accv[1], accv[2], accv[3]);
    if constexpr (FIN) asm volatile("" ::"v"(f_act[0]), "v"(f_act[1]), "v"(f_act[2]), "v"(f_act[3]), "v"(f_step[0]), "v"(f_step[1]), "v"(f_step[2]), "v"(f_step[3]));
    if constexpr (CHAIN) {   // the next step's noise row (the counters it hangs on were requested before the products)
        if (c_noise && fu.chain_noise && m0 + c_r < p_m && c_act && c_st + 1 < c_max)
            c_vs = *reinterpret_cast<const float2*>(fu.chain_noise + (int64_t)(m0 + c_r) * fu.chain_noise_stride + (int64_t)(c_st + 1) * SO_K + c_c);
    }
    __syncthreads();
    SK_STAMP(5);
    if constexpr (!CHAIN) {
        if (kq4 > 0) return;
        if (!n_ok) return;
    }
    __shared__ __attribute__((aligned(16))) unsigned char c_planes[CHAIN ? 4 * 16 * SO_PITCH : 16];   // frame hi / lo, noise hi / lo
    if (!CHAIN || (kq4 == 0 && n_ok)) {
    float acc[4] = {accv[0], accv[1], accv[2], accv[3]};
#pragma unroll
    for (int t = 1; t < KP; t++) {
        float4 p = red[((t - 1) * CG + cg) * 64 + lane];
        acc[0] += p.x; acc[1] += p.y; acc[2] += p.z; acc[3] += p.w;
    }
    if (STAMP) { if (acc[0] == 1.2345e-30f) SK_STAMP(6); SK_STAMP(1); }   // (stamp build) sums final; slots 1 / 2 are rewritten here: the epilogue's own phases
    // D layout of 16x16x32: column (n) = lane & 15, row (m) = (lane >> 4) * 4 + reg
    if (splitk > 1) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            int m = m0 + q * 4 + reg;
            float v = acc[reg] * e_ws;
            if (z == 0 && a.R) v = e_r[reg] + (v + e_bias);
            if (m < p_m && n_ok) partial[(int64_t)z * (a.zstride ? a.zstride : (int64_t)p_m * p_n) + (int64_t)m * p_n + n] = v;   // (zstride: the planes of a row CHUNK of a taller operand)
        }
        return;
    }
    // One pass over the epilogue form for the lane's four values (not the form re-decided per value: the unrolled switch was 1100
    // instructions of branches that every storing wave walked through cold -- in-situ stamps put 1.7-2.2 us between "sums final" and
    // "stores issued" in EVERY variant, two fifths of a small launch, profiles/r3_step_stamps_epilogue.txt).
    float v[4];
#pragma unroll
    for (int reg = 0; reg < 4; reg++) v[reg] = acc[reg] * e_ws + e_bias;
    const bool to_tail = a.tail != nullptr && n == p_n - 1;   // the out_eos column rides as the last column of cond_embed
    switch (a.epi) {
        case EPI_NONE: break;
        case EPI_GELU:
#pragma unroll
            for (int reg = 0; reg < 4; reg++) v[reg] = to_tail ? v[reg] : gelu1(v[reg]);
            break;
        case EPI_SILU:
#pragma unroll
            for (int reg = 0; reg < 4; reg++) v[reg] = to_tail ? v[reg] : silu1(e_addv + v[reg]);
            break;
        case EPI_ELU:
#pragma unroll
            for (int reg = 0; reg < 4; reg++) v[reg] = to_tail ? v[reg] : elu1(v[reg]);
            break;
        case EPI_RESADD:
#pragma unroll
            for (int reg = 0; reg < 4; reg++) v[reg] = to_tail ? v[reg] : e_r[reg] + v[reg];
            break;
        case EPI_SCALE_RESADD:
#pragma unroll
            for (int reg = 0; reg < 4; reg++) v[reg] = to_tail ? v[reg] : e_r[reg] + e_scl * v[reg];
            break;
        case EPI_GATE_RESADD:
#pragma unroll
            for (int reg = 0; reg < 4; reg++) v[reg] = to_tail ? v[reg] : e_r[reg] + e_g[reg] * v[reg];
            break;
        case EPI_AXPY:
#pragma unroll
            for (int reg = 0; reg < 4; reg++) v[reg] = to_tail ? v[reg] : e_r[reg] + a.alpha * v[reg];
            break;
        case EPI_RESADD_ELU:
#pragma unroll
            for (int reg = 0; reg < 4; reg++) v[reg] = to_tail ? v[reg] : elu1(e_r[reg] + v[reg]);
            break;
    }
    float* const cbase = CHAIN ? nullptr : (to_tail ? a.tail : a.C);   // chained: C (the Euler state) receives the next step's x0 below, the frame goes to the latents
    const int64_t cld = to_tail ? 1 : a.cmap.ld;
    const int ccol = to_tail ? 0 : n;
    if constexpr (CHAIN) {   // the frame as the next step's input: NaN -> bos (tensor_util.go:259-268), bf16 hi / lo planes [16][32]
#pragma unroll
        for (int reg = 0; reg < 4; reg++) so_put(c_planes, c_planes + 16 * SO_PITCH, q * 4 + reg, n, isnan(v[reg]) ? c_bos : v[reg]);
    }
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int m = m0 + q * 4 + reg;
        if (m < p_m && cbase) cbase[(int64_t)m * cld + ccol] = v[reg];
        if constexpr (FIN)
            if (m < p_m && !to_tail && f_act[reg] && (!CHAIN || blockIdx.x == 0)) fu.fin->latents[(int64_t)m * fu.fin->lat_stride + (int64_t)f_step[reg] * fu.fin->ldim + n] = v[reg];   // latentFrames = append(...)
    }
    if constexpr (FIN)
    if (cg == 0 && (lane & 15) == 0 && (!CHAIN || blockIdx.x == 0)) {   // k_step_finish's bookkeeping (runtime_native_safetensors.go:176-192), one lane per row
        const StepState& fs = fu.fin->s;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int m = m0 + q * 4 + reg;
            if (m >= p_m || !f_act[reg]) continue;
            const int st = f_step[reg];
            const bool is_eos = f_logit[reg] > f_thr[reg];   // flow_lm.go:281
            int cd = f_cd[reg];
            bool done = false;
            if (is_eos && cd < 0) { cd = f_fae[reg]; fs.eos_step[m] = st; }
            if (cd >= 0) {
                if (cd == 0) { done = true; fs.broke[m] = 1; }
                else cd--;
            }
            fs.countdown[m] = cd;
            fs.n_frames[m] = st + 1;
            fs.step[m] = st + 1;
            fs.kv_len[m] = f_kv[reg] + 1;
            if (st + 1 >= f_max[reg]) done = true;
            if (done) { fs.active[m] = 0; atomicSub(fs.n_active, 1); }
        }
    }
    if (STAMP) { SK_STAMP(2); __builtin_amdgcn_s_waitcnt(0); SK_STAMP(6); }   // stores issued; stores acknowledged
    }   // (the storing waves)
    if constexpr (CHAIN) {
        unsigned char* const fh = c_planes, * const fl = c_planes + 16 * SO_PITCH, * const nh = c_planes + 32 * SO_PITCH, * const nl = c_planes + 48 * SO_PITCH;
        if (c_noise) {
            so_put(nh, nl, c_r, c_c, c_vs.x);
            so_put(nh, nl, c_r, c_c + 1, c_vs.y);
        }
        __syncthreads();   // planes complete; every read of this step's x0 (the Euler update's residual) has been consumed
        if (c_noise && m0 + c_r < p_m) *reinterpret_cast<float2*>(c_x0 + (int64_t)(m0 + c_r) * SO_K + c_c) = c_vs;
        const unsigned char* const ph = c_fx ? nh : fh, * const pl = c_fx ? nl : fl;
        // block 0: every column of fx; block j > 0: columns [(j-1) 512, j 512) of x, then -- a launch with fewer blocks than x has slices -- every gridDim.x-1-th slice after it
        const int stride_t = c_fx ? 16 * CT : ((int)gridDim.x - 1) * 16 * CT;
        for (int tb = c_t0; tb * 16 < c_n; tb += stride_t) {   // (one pass at the model's widths)
#pragma unroll
            for (int t = 0; t < CT; t++) {
                const int n0 = (tb + wave * CT + t) * 16;
                if (n0 >= c_n) break;
                const SoW<CWB> w = tb == c_t0 ? cw[t] : so_load_w<CWB>(c_wp, c_bp, n0, c_n, lane);
                so_tile<CWB>(ph, pl, w, n0, c_out, c_n, m0, p_m, lane);
            }
        }
    }
#undef SK_STAMP
}

bool skinny_supported(const GemmArgs& a, int splitk) {
    if (splitk < 1) splitk = 1;
    const int kslice = splitk > 1 ? ((a.K + splitk - 1) / splitk + 127) / 128 * 128 : a.K;
    return a.Wt && a.M <= kStepMaxRows && a.K % 8 == 0 && kslice <= (splitk > 1 && (a.w_bf16 || a.wt_i8) ? SK_KMAX2 : SK_KMAX) && (!a.wt_i8 || a.wscale) && a.amap.rows_per_batch == 0 && a.cmap.rows_per_batch == 0 &&
           a.amap.ld % 4 == 0 && aligned16(a.A) && aligned16(a.W) && a.ldw % 8 == 0 && a.aop == AOP_NONE;
}

bool skinny_fuse_supported(const GemmArgs& a, const SkinnyFuse& f) {
    // the fused prologue needs whole rows in one block: K is the row width, one K slice, dense rows
    return skinny_supported(a, 1) && a.K <= SK_KMAX && a.amap.ld == a.K && a.K % 4 == 0 && (!f.scale || f.ldmod % 4 == 0) &&
           !f.pgate && (!f.ln_w == !f.ln_b) && (!f.partial || f.psplit >= 1) && (f.ln || !(f.ln_w || f.scale)) &&
           (!f.fin || (a.N <= 64 && a.K <= 512 && f.ln && f.scale && !f.ln_w && !f.partial)) &&   // fin: instantiated for the flow net's final layer only
           (!f.chain || (f.fin && a.N == SO_K && a.epi == EPI_AXPY && !a.tail && f.chain_noise_stride % 2 == 0));   // chain: the frame is the whole row of C (ldim == 32)
}

thread_local hipEvent_t g_skinny_ev[2] = {nullptr, nullptr};   // measurement pass (bench.py roofline): the kernel's own begin / end timestamps
thread_local unsigned long long* g_skinny_stamps = nullptr;   // debug (ptts_debug_skinny_stamps)
thread_local SkinnyStampLog* g_skinny_stamp_log = nullptr;    // debug (ptts_debug_step_stamps)

template <int WBF16, int PRO, int NJ, int CG>
static void launch_cg(const GemmArgs& a, const SkinnyFuse& fu, int splitk, float* partial, hipStream_t stream) {
    dim3 grid((a.N + 16 * CG - 1) / (16 * CG), (a.M + 15) / 16, splitk);
    if constexpr (PRO == (PRO_LN | PRO_MOD) && NJ == 2 && CG == 4) {
        if (fu.fin && fu.chain) {   // ... which also opens the next step: SK_CHAIN_BX blocks per row tile (k_skinny, `cbx`)
            grid.x = SK_CHAIN_BX;
            if (g_skinny_ev[0])
                hipExtLaunchKernelGGL((k_skinny<WBF16, false, PRO, NJ, CG, true, true>), grid, dim3(1024), 0, stream, g_skinny_ev[0], g_skinny_ev[1], 0, a.Wt, a.A, (int)a.amap.ld,
                                      a.M | (splitk << 16), a.N, a.K, fu.partial, fu.ln_w, fu.ln_b, a, fu, partial, (unsigned long long*)nullptr);
            else hipLaunchKernelGGL((k_skinny<WBF16, false, PRO, NJ, CG, true, true>), grid, dim3(1024), 0, stream, a.Wt, a.A, (int)a.amap.ld, a.M | (splitk << 16), a.N, a.K, fu.partial, fu.ln_w, fu.ln_b, a, fu, partial,
                                    (unsigned long long*)nullptr);
            return;
        }
        if (fu.fin) {   // the flow net's final layer with the step's bookkeeping in its epilogue (one column block: grid.x == 1)
            if (g_skinny_ev[0])
                hipExtLaunchKernelGGL((k_skinny<WBF16, false, PRO, NJ, CG, true>), grid, dim3(1024), 0, stream, g_skinny_ev[0], g_skinny_ev[1], 0, a.Wt, a.A, (int)a.amap.ld,
                                      a.M | (splitk << 16), a.N, a.K, fu.partial, fu.ln_w, fu.ln_b, a, fu, partial, (unsigned long long*)nullptr);
            else hipLaunchKernelGGL((k_skinny<WBF16, false, PRO, NJ, CG, true>), grid, dim3(1024), 0, stream, a.Wt, a.A, (int)a.amap.ld, a.M | (splitk << 16), a.N, a.K, fu.partial, fu.ln_w, fu.ln_b, a, fu, partial,
                                    (unsigned long long*)nullptr);
            return;
        }
    }
    if (SkinnyStampLog* lg = g_skinny_stamp_log) {
        const size_t blocks = (size_t)grid.x * grid.y * grid.z;
        if (lg->used_blocks + blocks <= lg->cap_blocks) {
            hipLaunchKernelGGL((k_skinny<WBF16, true, PRO, NJ, CG>), grid, dim3(1024), 0, stream, a.Wt, a.A, (int)a.amap.ld, a.M | (splitk << 16), a.N, a.K, fu.partial, fu.ln_w, fu.ln_b, a, fu, partial,
                               lg->base + 8 * lg->used_blocks);
            lg->used_blocks += blocks;
            lg->desc.push_back(SkinnyStampLog::Desc{a.M, a.N, a.K, PRO, NJ, CG, (int32_t)blocks, splitk});
            return;
        }
    }
    if (g_skinny_stamps) hipLaunchKernelGGL((k_skinny<WBF16, true, PRO, NJ, CG>), grid, dim3(1024), 0, stream, a.Wt, a.A, (int)a.amap.ld, a.M | (splitk << 16), a.N, a.K, fu.partial, fu.ln_w, fu.ln_b, a, fu, partial, g_skinny_stamps);
    else if (g_skinny_ev[0])   // hipExtLaunchKernel stamps the dispatch itself: the same interval rocprofv3 reports for the kernel
        hipExtLaunchKernelGGL((k_skinny<WBF16, false, PRO, NJ, CG>), grid, dim3(1024), 0, stream, g_skinny_ev[0], g_skinny_ev[1], 0, a.Wt, a.A, (int)a.amap.ld, a.M | (splitk << 16), a.N,
                              a.K, fu.partial, fu.ln_w, fu.ln_b, a, fu, partial, (unsigned long long*)nullptr);
    else hipLaunchKernelGGL((k_skinny<WBF16, false, PRO, NJ, CG>), grid, dim3(1024), 0, stream, a.Wt, a.A, (int)a.amap.ld, a.M | (splitk << 16), a.N, a.K, fu.partial, fu.ln_w, fu.ln_b, a, fu, partial, (unsigned long long*)nullptr);
}

template <int WBF16, int PRO, int NJ>
static void launch_nj(const GemmArgs& a, const SkinnyFuse& fu, int splitk, float* partial, dim3 /*grid*/, hipStream_t stream) {
    // narrow blocks when the 64-column grid would occupy fewer than half of the 256 CUs
    const int blocks64 = ((a.N + 63) / 64) * ((a.M + 15) / 16) * splitk;
    // ... unless, beyond 64 rows, the narrow grid then needs a second round of CUs (513 columns x 128 rows: 264 blocks).  Up to 64 rows the choice stays what it
    // was: the K parts a wave sums (the rounding order) follow it, and a continuous engine's packed prefill must give a prompt the bits its stand-alone prefill gives
    const int blocks16 = ((a.N + 15) / 16) * ((a.M + 15) / 16) * splitk;
    if (blocks64 < 128 && (a.M <= 64 || blocks16 <= 256) && a.N > 16 && !fu.fin) launch_cg<WBF16, PRO, NJ, 1>(a, fu, splitk, partial, stream);   // fin: one column block (see SkinnyFuse)
    else launch_cg<WBF16, PRO, NJ, 4>(a, fu, splitk, partial, stream);
}

template <int WBF16, int PRO>
static void launch_pro(const GemmArgs& a, const SkinnyFuse& fu, int splitk, float* partial, dim3 grid, hipStream_t stream) {
    const int kslice = splitk > 1 ? ((a.K + splitk - 1) / splitk + 127) / 128 * 128 : a.K;
    if (kslice <= 512) launch_nj<WBF16, PRO, 2>(a, fu, splitk, partial, grid, stream);
    else if (kslice <= SK_KMAX) launch_nj<WBF16, PRO, 4>(a, fu, splitk, partial, grid, stream);
    else if constexpr (PRO == 0 && WBF16 != 0) launch_cg<WBF16, 0, 8, 2>(a, fu, splitk, partial, stream);   // 2048-deep slices: 32-column blocks x 8 K parts
    else throw Error(PTTS_EINVAL, "ptts-hip: internal: K slice too deep for this variant of the step kernel");
}

template <int WBF16>
static void launch_w(const GemmArgs& a, const SkinnyFuse& fu, int splitk, float* partial, dim3 grid, hipStream_t stream) {
    const int pro = (fu.ln ? PRO_LN : 0) | ((fu.ln && fu.ln_w) ? PRO_AFFINE : 0) | ((fu.ln && fu.scale) ? PRO_MOD : 0) | (fu.partial ? PRO_PARTIAL : 0);
    switch (pro) {
        case 0: launch_pro<WBF16, 0>(a, fu, splitk, partial, grid, stream); break;
        case PRO_LN | PRO_AFFINE: launch_pro<WBF16, PRO_LN | PRO_AFFINE>(a, fu, splitk, partial, grid, stream); break;
        case PRO_LN | PRO_AFFINE | PRO_PARTIAL:
            if (fu.psplit == 1) launch_pro<WBF16, PRO_LN | PRO_AFFINE | PRO_PARTIAL | PRO_ONE>(a, fu, splitk, partial, grid, stream);
            else launch_pro<WBF16, PRO_LN | PRO_AFFINE | PRO_PARTIAL>(a, fu, splitk, partial, grid, stream);
            break;
        case PRO_LN | PRO_AFFINE | PRO_MOD: launch_pro<WBF16, PRO_LN | PRO_AFFINE | PRO_MOD>(a, fu, splitk, partial, grid, stream); break;
        case PRO_LN | PRO_MOD: launch_pro<WBF16, PRO_LN | PRO_MOD>(a, fu, splitk, partial, grid, stream); break;
        case PRO_LN: launch_pro<WBF16, PRO_LN>(a, fu, splitk, partial, grid, stream); break;
        case PRO_PARTIAL: launch_pro<WBF16, PRO_PARTIAL>(a, fu, splitk, partial, grid, stream); break;   // split-K sum + residual, no norm
        default: launch_pro<WBF16, PRO_LN | PRO_AFFINE | PRO_MOD | PRO_PARTIAL>(a, fu, splitk, partial, grid, stream); break;
    }
}

void launch_skinny(const GemmArgs& a, const SkinnyFuse& fu, int splitk, float* partial, hipStream_t stream) {
    if (a.M <= 0 || a.N <= 0) return;
    note_launch("k_skinny");
    dim3 grid((a.N + 63) / 64, (a.M + 15) / 16, splitk);
    if (a.wt_i8) launch_w<2>(a, fu, splitk, partial, grid, stream);
    else if (a.w_bf16) launch_w<1>(a, fu, splitk, partial, grid, stream);
    else launch_w<0>(a, fu, splitk, partial, grid, stream);
}

}  // namespace ptts
