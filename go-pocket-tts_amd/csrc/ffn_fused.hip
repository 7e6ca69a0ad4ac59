// ffn_fused.hip -- the feed-forward half of a Mimi decoder-transformer layer as ONE kernel (mimi.go:245-285,351-358, 506-525):
//     x += layer_scale_2 * linear2( gelu( linear1( LayerNorm2(x) ) ) )        x: [rows][512] f32, updated in place
// Before: k_layernorm_reg -> k_gemm_wres (linear1 + GELU, K = 512, N = 2048) -> k_gemm5 (linear2, K = 2048, N = 512): the normalised rows
// (262 MB per layer at batch 64 x 10 s) and the 2048-wide hidden rows (1.05 GB) were written to HBM and read back.  Here neither exists in
// memory: a wave owns 16 rows for the whole layer half, and everything between its rows and its rows stays in its registers.
//
//   * block = 4 waves (one per SIMD, up to 512 registers each) = 64 rows; the grid covers the rows.
//   * prologue: the wave loads its 16 rows (full 128-byte lines), LayerNorm in f32 (biased variance, linear.go:295-309), and keeps the result as
//     the B operand of v_mfma_f32_16x16x32_bf16 for all 16 k steps, split into bf16 hi + lo (x = hi + lo to 2^-17): 128 registers.
//   * the hidden width is walked in chunks of 32 units.  Per chunk:  H^T[32 x 16 rows] = W1[chunk] x X^T  (32 MFMA pairs, weights as the A operand),
//     GELU(erf) on the 8 sums a lane holds, split into hi + lo -- and those 8 values ARE the B operand of the second product in the k order
//     its weights were laid out for (lane (row, q): hidden units 4q..4q+3 and 16+4q..16+4q+3 of the chunk):  Y^T[512 x 16 rows] += W2[:, chunk] x H^T
//     (32 MFMA pairs into 128 accumulator registers).  bf16 weights are exact, so every product is exact to the f32 rounding of the splits, sums in f32.
//   * weights reach the matrix cores through LDS: per chunk a 32-KB image of W1's rows and a 32-KB image of W2's columns, prepared at load time in
//     exactly the layout the fragment reads want (model.cpp add_ffn_image: bank-conflict-free for ds_read_b128, k order as above), so a stage is a
//     linear global -> LDS copy by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write).  Two W1 slots + two W2 slots (128 KB); the
//     chunk loop is software-pipelined (first product of chunk c+1, then the second product of chunk c), one barrier per chunk, the copies of a
//     chunk are issued a whole iteration before they are needed.
//   * epilogue: x + scale * y, written over the rows the wave read (nobody else touches them).
// Cost model per layer at 128 000 rows: 8192 MFMA per wave-tile x 16 cycles -> 0.51 ms at 2 GHz if the matrix cores never waited; 64 KB of weight
// images per chunk per CU = 57 GB/s per CU, under the ~70 GB/s a CU takes in from its XCD's L2.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "device_util.h"

namespace ptts {

namespace {

typedef __bf16 ff_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ff_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ff_f32x2 __attribute__((ext_vector_type(2)));
typedef float ff_f32x4 __attribute__((ext_vector_type(4)));

union FfFrag {
    ff_bf16x8 v;
    uint4 q;
    unsigned u[4];
};

__device__ __forceinline__ void ff_split2(float a, float b, unsigned& hi, unsigned& lo) {
    ff_f32x2 f = {a, b};
    ff_bf16x2 h = __builtin_convertvector(f, ff_bf16x2);
    ff_f32x2 r = f - __builtin_convertvector(h, ff_f32x2);
    ff_bf16x2 l = __builtin_convertvector(r, ff_bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

// GELU(erf) with erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute): 1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z = |x| / sqrt 2.  Branch-free,
// ~16 vector instructions with two transcendentals (v_rcp_f32, v_exp_f32) -- libm's erff is ~40 with a divergent branch, and this kernel evaluates it
// 8 times per lane between two runs of matrix instructions.  Measured against f64 on 2e6 points of [-8, 8]: |gelu error| <= 4.7e-7 (at x = 3.1), the level
// of 0.5 x (1 + erff(x / sqrt 2)) itself, whose 1 + erf cancels the same way for negative x.
__device__ __forceinline__ float ff_gelu(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));   // argument of erf: |x| / sqrt 2
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * 0.5f * x * x);   // exp(-(x / sqrt 2)^2)
    const float erfa = fmaf(-p, e, 1.0f);                                          // erf(|x| / sqrt 2)
    const float erfs = copysignf(erfa, x);
    return 0.5f * x * (1.0f + erfs);
}

constexpr int FF_D = 512;                 // model width (rows of x, K of linear1, N of linear2)
constexpr int FF_CH = 32;                 // hidden units per chunk
constexpr int FF_KS = FF_D / 32;          // k steps of the first product (16)
constexpr int FF_OT = FF_D / 16;          // output tiles of the second product (32)
constexpr int FF_W1B = FF_CH * FF_D * 2;  // bytes of a W1 chunk image (32 KB)
constexpr int FF_W2B = FF_D * FF_CH * 2;  // bytes of a W2 chunk image (32 KB)
constexpr int FF_IMG = FF_W1B + FF_W2B;   // one chunk's images, contiguous in the arena

// 32 KB global -> LDS by the block's four waves: 32 wave-instructions of 1 KB (lane l: 16 bytes at src + 16 l -> dst + 16 l).
// Inline assembly on purpose: told about an LDS-DMA write (the builtin), hipcc puts `s_waitcnt vmcnt(0)` in front of the next ds_read of ANY slot --
// it cannot tell the slots apart -- which is the copy waited for at once instead of an iteration later.  So the copies are invisible to its
// counters and ordered by hand: every wave waits vmcnt(0) at the one place below where a slot changes hands, in front of the block's barrier.
// (The compiler's own vmcnt(N) waits stay right: loads return in issue order, so a wait that lets the N youngest VISIBLE loads stay in flight
// has also seen every older copy land; and no visible load is issued between a copy and that wait.)  m0 carries the LDS address; nothing else
// in this kernel uses it -- and the compiler keeps nothing in it across statements: m0 is a RESERVED register to LLVM (listing it as a clobber is rejected as
// "reserved registers on the clobber list"), written right in front of each of the compiler's own uses (s_sendmsg, GWS, v_movrel), none of which these kernels have.
// One LDS-DMA wave-instruction: lane l's 16 bytes at src_lane -> dst + 16 l.  The wait state between the write of m0 and its use is written out because
// the compiler's hazard recognizer does not look inside inline assembly (ISA: SALU write of M0 -> LDS "direct" / GDS / sendmsg use needs one).
__device__ __forceinline__ void ff_dma1k(const char* src_lane, char* dst) {
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(base), "v"(src_lane) : "memory");
}
__device__ __forceinline__ void ff_dma32k(const char* src, char* dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int idx = i * 4 + wave;
        ff_dma1k(src + idx * 1024 + lane * 16, dst + idx * 1024);
    }
}

// The wave's 16 rows (lane (row, q): its quarter of one row) -> LayerNorm in f32 (biased variance, linear.go:295-309) -> the B operand of
// v_mfma_f32_16x16x32_bf16 for the 16 k steps, bf16 hi + lo (x = hi + lo to 2^-17).  k step s, lane (row, q): k = 32 s + 4 q + (0..3) and
// 32 s + 16 + 4 q + (0..3) -- the k order the weight images are laid out in (and the order in which a pair of 16x16 accumulator tiles would hand
// its values over: these kernels can take their rows from a product later).
__device__ __forceinline__ void ff_rows_layernorm(const float* xrow, const float* ln_w, const float* ln_b, float eps, int q, FfFrag (&xh)[16], FfFrag (&xl)[16]) {
    float4 v[16][2];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        v[s][0] = *reinterpret_cast<const float4*>(xrow + 32 * s + 4 * q);
        v[s][1] = *reinterpret_cast<const float4*>(xrow + 32 * s + 16 + 4 * q);
    }
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < 16; s++) sum += (v[s][0].x + v[s][0].y) + (v[s][0].z + v[s][0].w) + (v[s][1].x + v[s][1].y) + (v[s][1].z + v[s][1].w);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / 512);
    float var = 0.f;
#pragma unroll
    for (int s = 0; s < 16; s++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float d0 = v[s][h].x - mean, d1 = v[s][h].y - mean, d2 = v[s][h].z - mean, d3 = v[s][h].w - mean;
            var += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    var += __shfl_xor(var, 16, 64);
    var += __shfl_xor(var, 32, 64);
    const float rstd = 1.0f / sqrtf(var * (1.0f / 512) + eps);
#pragma unroll
    for (int s = 0; s < 16; s++) {
        float4 o[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int k = 32 * s + 16 * h + 4 * q;
            const float4 g = *reinterpret_cast<const float4*>(ln_w + k), b = *reinterpret_cast<const float4*>(ln_b + k);
            o[h].x = (v[s][h].x - mean) * rstd * g.x + b.x;
            o[h].y = (v[s][h].y - mean) * rstd * g.y + b.y;
            o[h].z = (v[s][h].z - mean) * rstd * g.z + b.z;
            o[h].w = (v[s][h].w - mean) * rstd * g.w + b.w;
        }
        ff_split2(o[0].x, o[0].y, xh[s].u[0], xl[s].u[0]);
        ff_split2(o[0].z, o[0].w, xh[s].u[1], xl[s].u[1]);
        ff_split2(o[1].x, o[1].y, xh[s].u[2], xl[s].u[2]);
        ff_split2(o[1].z, o[1].w, xh[s].u[3], xl[s].u[3]);
    }
}

// First product of one 32-row weight chunk against the wave's rows: acc[ht] (ht = 0, 1) = W[chunk rows 16 ht ..][512] x X^T, 16 k steps x (two fragment
// reads, four matrix instructions) from a W1-format image in LDS (model.cpp add_w1_image).  The fragments of step s+1 are requested before step s is
// multiplied; piece(s) is other work that rides in step s's shadow.  Each step is fenced (sched_barrier) and its order pinned (sched_group_barrier) --
// left to itself the scheduler issues read, wait, multiply back to back and piles the vector work up in front of the matrix instructions.
template <class Piece>
__device__ __forceinline__ void ff_gemm1(const char* w1, int r16, int q, const FfFrag (&xh)[16], const FfFrag (&xl)[16], ff_f32x4 (&acc)[2], Piece&& piece) {
    const int sw1 = (r16 >> 1) & 7;
    acc[0] = ff_f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = ff_f32x4{0.f, 0.f, 0.f, 0.f};
    FfFrag wa[2], wb[2];
    const char* wl = w1 + r16 * 128;
    wa[0].q = *reinterpret_cast<const uint4*>(wl + ((q ^ sw1) << 4));
    wb[0].q = *reinterpret_cast<const uint4*>(wl + ((q ^ sw1) << 4) + 2048);
#pragma unroll
    for (int s = 0; s < 16; s++) {
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < 16) {
            const int off = ((s + 1) >> 1) * 4096 + (((4 * ((s + 1) & 1) + q) ^ sw1) << 4);
            wa[(s + 1) & 1].q = *reinterpret_cast<const uint4*>(wl + off);
            wb[(s + 1) & 1].q = *reinterpret_cast<const uint4*>(wl + off + 2048);
        }
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s & 1].v, xh[s].v, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[s & 1].v, xh[s].v, acc[1], 0, 0, 0);
        PTTS_LO_MFMA(acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s & 1].v, xl[s].v, acc[0], 0, 0, 0));
        PTTS_LO_MFMA(acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[s & 1].v, xl[s].v, acc[1], 0, 0, 0));
        piece(s);
        // (both accumulator chains are made to exist HERE: matrix instructions are pure values to the instruction selector, which otherwise lets the
        // second chain float to the end of the block -- behind whatever follows -- before the scheduler's fences ever see it)
        asm volatile("" : "+a"(acc[0]), "+a"(acc[1]));
        if (s + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // (Round 4 fenced the end of this chain with 32 idle states and tied every reader of the sums to a later point, after k_mimi_rowlin's first cut had returned
    // one output column -- lanes 48..63 of one register -- without a product.  Round 5 found the cause with the failing cut rebuilt and bisected in its ISA
    // (tools/probes/mfma_hazard/README.md): not the accumulators at all, but the RoPE rotation behind them -- hipcc's SLP vectoriser had packed v[2] * c.y into
    // `v_pk_mul_f32 ... op_sel:[0,1]`, the scheduler had put the next matrix instruction right behind it, and on gfx950 a packed-f32 op whose low result mixes the
    // dword halves of its sources returns a wrong low result in its last pass (lanes 48..63) when it is issued in the shadow of one MFMA and followed at once by
    // another.  This file is therefore compiled with -fno-slp-vectorize (Makefile), tools/isa_hazard_check.py scans every shipped code object for the pattern
    // (tests/test_isa_hazards.py), and the idle states are gone.)
}

}  // namespace

// V (measurement): 0 the product; 1 no weight copies inside the loop (the first chunk's images stay: what the copies cost).  (Copies through registers --
// global_load_dwordx4 -> ds_write_b128, 64 staging registers -- were measured too: 4230 us per layer, not kept.)
template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_mimi_ffn(FfnArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[2 * FF_W1B + 2 * FF_W2B];   // W1 slots at 0 and 32 K, W2 slots at 64 K and 96 K
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int nch = a.F / FF_CH;
    const char* img = reinterpret_cast<const char*>(a.img);
    auto w1s = [&](int i) { return lds + i * FF_W1B; };
    auto w2s = [&](int i) { return lds + 2 * FF_W1B + i * FF_W2B; };

    // the first three copies: W1 of chunks 0 and 1, W2 of chunk 0 (nothing depends on the rows)
    ff_dma32k(img, w1s(0), wave, lane);
    ff_dma32k(img + FF_W1B, w2s(0), wave, lane);
    ff_dma32k(img + (size_t)min(1, nch - 1) * FF_IMG, w1s(1), wave, lane);

    // ---- the wave's 16 rows -> LayerNorm -> bf16 hi / lo B-operand fragments for the 16 k steps ----
    const int m0 = blockIdx.x * 64 + wave * 16;
    const int row = min(m0 + r16, a.M - 1);
    const float* xrow = a.x + row_off(a.xmap, row);
    FfFrag xh[FF_KS], xl[FF_KS];
    ff_rows_layernorm(xrow, a.ln_w, a.ln_b, a.eps, q, xh, xl);

    ff_f32x4 acc2[FF_OT];
#pragma unroll
    for (int t = 0; t < FF_OT; t++) acc2[t] = ff_f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses inside a slot.  W1 image: [k pair j (8)][hidden unit h (32)][128 B], the 16-byte chunk (k-step parity sg, lane group q) of
    // row h at position (4 sg + q) ^ ((h >> 1) & 7); W2 image: [output o (512)][64 B], lane group q's chunk at q ^ (3 * ((o >> 3) & 1)).
    const int w2_lane = r16 * 64 + ((q ^ (((r16 >> 3) & 1) * 3)) << 4);   // + t * 1024

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the first copies has landed
    __syncthreads();                                   // ... and everybody else's
    ff_f32x4 acc1[2], acc1n[2];
    ff_gemm1(w1s(0), r16, q, xh, xl, acc1, [](int) {});

    for (int c = 0; c < nch; c++) {
        // every wave has finished iteration c-1 (its first product read W1 slot c&1 -- chunk c --, its second W2 slot (c-1)&1) and the copies issued
        // during it (W1 of chunk c+1, W2 of chunk c) have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        asm volatile("" : "+a"(acc1[0]), "+a"(acc1[1]));   // (no read of the previous chunk's sums is scheduled in front of the barrier: note at the end of ff_gemm1)
        // the copies of this iteration: W1 of chunk c+2 into the slot chunk c's first product has left, W2 of chunk c+1 into the slot chunk c-1's second
        // product has left -- 16 wave-instructions per wave, ONE PER K STEP of the first product below.  Issued in a burst behind the barrier (the first
        // cut) the four waves' 64 KB met in the CU's one vector-memory path and every wave stood in front of its matrix instructions until its own
        // sixteen were accepted: 1263 us per layer against 1024 without any copies.
        const char* const src1 = img + (size_t)min(c + 2, nch - 1) * FF_IMG + wave * 1024 + lane * 16;
        const char* const src2 = img + (size_t)min(c + 1, nch - 1) * FF_IMG + FF_W1B + wave * 1024 + lane * 16;
        char* const dst1 = w1s(c & 1) + wave * 1024;
        char* const dst2 = w2s((c + 1) & 1) + wave * 1024;
        // (past the last chunk the last one is copied again, into a slot nobody reads any more: no branch inside the fenced steps)
        // this chunk's activation in the shadow of the next chunk's first product: GELU(erf) (tensor_util.go:84-94) on the 8 sums the lane holds (one per
        // two k steps), split into hi + lo (one pair per four steps) -> the second product's B operand
        FfFrag hh, hl;
        float g[8];
        ff_gemm1(w1s(min(c + 1, nch - 1) & 1), r16, q, xh, xl, acc1n, [&](int s) {   // (past the last chunk: the last slot once more, never used -- no branch around 64 matrix instructions)
            const int i = s >> 1;
            if ((s & 1) == 0) g[i] = ff_gelu(acc1[i >> 2][i & 3]);
            else if (i & 1) ff_split2(g[i - 1], g[i], hh.u[i >> 1], hl.u[i >> 1]);
            if constexpr (V == 0) {
                if (s < 8) ff_dma1k(src1 + s * 4096, dst1 + s * 4096);
                else ff_dma1k(src2 + (s - 8) * 4096, dst2 + (s - 8) * 4096);
            }
        });
        // second product: Y^T += W2[:, chunk] x H^T, 32 output tiles x (one fragment read, two matrix instructions), fragment t+1 requested before t is multiplied
        const char* w2 = w2s(c & 1) + w2_lane;
        FfFrag w2f[2];
        w2f[0].q = *reinterpret_cast<const uint4*>(w2);
#pragma unroll
        for (int t = 0; t < FF_OT; t++) {
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < FF_OT) w2f[(t + 1) & 1].q = *reinterpret_cast<const uint4*>(w2 + (t + 1) * 1024);
            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t & 1].v, hh.v, acc2[t], 0, 0, 0);
            PTTS_LO_MFMA(acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t & 1].v, hl.v, acc2[t], 0, 0, 0));
            if (t + 1 < FF_OT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc1[0] = acc1n[0];
        acc1[1] = acc1n[1];
    }

    // ---- epilogue: lane (row, q) holds y[row][16 t + 4 q .. + 3]; out = x + scale * y over the rows that were read ----
    if (m0 + r16 < a.M) {
        float* orow = a.x + row_off(a.xmap, m0 + r16);
        const bool has_ls = a.ls != nullptr;
        const float* lsp = has_ls ? a.ls : a.ln_w;   // (no scale: any readable address, the value is discarded -- no load behind a branch)
#pragma unroll
        for (int t0 = 0; t0 < FF_OT; t0 += 8) {
            float4 r[8], sc[8];
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int col = 16 * (t0 + t) + 4 * q;
                r[t] = *reinterpret_cast<const float4*>(orow + col);
                sc[t] = *reinterpret_cast<const float4*>(lsp + col);
                if (!has_ls) sc[t] = make_float4(1.f, 1.f, 1.f, 1.f);
            }
#pragma unroll
            for (int t = 0; t < 8; t++) {
#pragma clang fp contract(off)
                const ff_f32x4 y = acc2[t0 + t];
                float4 o;
                o.x = r[t].x + sc[t].x * y[0];
                o.y = r[t].y + sc[t].y * y[1];
                o.z = r[t].z + sc[t].z * y[2];
                o.w = r[t].w + sc[t].w * y[3];
                *reinterpret_cast<float4*>(orow + 16 * (t0 + t) + 4 * q) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm + linear (+ RoPE on the leading columns) with the rows resident in registers: the Mimi transformer's norm1 + in_proj + q / k rotation
// (mimi.go:245-441: LN -> QKV without bias -> interleaved-pair RoPE at positions 0..T-1, rope.go:81-105).
// Before: k_layernorm_reg (rows written and read back: 2 x 262 MB per layer) -> k_gemm5 + RoPE epilogue, whose six column tiles per row panel fetched
// the panel 2.5 times.  Here a wave normalises its 16 rows once (ff_rows_layernorm), keeps them as matrix operands, and the [N][512] weights stream
// past them through LDS in chunks of 32 output columns (a 32-KB W1-format image each, three slots, LDS-DMA one k step apart); a chunk's 2 x 4 sums per
// lane are rotated and stored (16-byte pieces, 128 contiguous bytes per row and chunk) in the shadow of the next chunk's matrix instructions.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_mimi_rowlin(RowLinArgs a) {
    constexpr int NS = 3;
    __shared__ __attribute__((aligned(16))) char lds[NS * FF_W1B];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int nch = a.N / FF_CH;   // even (host)
    const char* img = reinterpret_cast<const char*>(a.img);
    ff_dma32k(img, lds, wave, lane);
    ff_dma32k(img + (size_t)min(1, nch - 1) * FF_W1B, lds + FF_W1B, wave, lane);

    const int m0 = blockIdx.x * 64 + wave * 16;
    const int row = min(m0 + r16, a.M - 1);
    FfFrag xh[FF_KS], xl[FF_KS];
    ff_rows_layernorm(a.x + row_off(a.xmap, row), a.ln_w, a.ln_b, a.eps, q, xh, xl);
    // RoPE: the lane's row sits at one position; its output columns 32 c + 16 ht + 4 q + (0..3) are two (even, odd) pairs of a 64-wide head, pair index
    // ((32 (c & 1) + 16 ht + 4 q) >> 1) + (0, 1): eight table entries per lane for the whole kernel
    float2 cs[4], sn[4];   // [2 (c & 1) + ht]: the two pairs' cos / sin
    const bool rope = a.rope_cos != nullptr;
    {
        const int pos = a.rope_pos0 + (a.rope_rows_per_seg ? row % a.rope_rows_per_seg : row);
        const float* ct = rope ? a.rope_cos + (int64_t)pos * 32 : a.ln_w;   // (no RoPE: any readable address, the values are not used)
        const float* st = rope ? a.rope_sin + (int64_t)pos * 32 : a.ln_w;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int jj = (32 * (i >> 1) + 16 * (i & 1) + 4 * q) >> 1;
            cs[i] = *reinterpret_cast<const float2*>(ct + jj);
            sn[i] = *reinterpret_cast<const float2*>(st + jj);
        }
    }
    float* const yrow = a.y + row_off(a.ymap, row);

    // rotate + store the two tiles of chunk cc (PAR = cc & 1, compile time: it selects the table registers), pieces p = 0 (tile 0), 1 (tile 1)
    auto store_piece = [&](ff_f32x4 (&acc)[2], int cc, auto par, int ht) {
#pragma clang fp contract(off)
        constexpr int PAR = decltype(par)::value;
        // (the reads of the sums stay HERE, five k steps and a barrier behind the matrix instruction that wrote them: see the note at the end of ff_gemm1)
        asm volatile("" : "+a"(acc[0]), "+a"(acc[1]));
        const int col = 32 * cc + 16 * ht + 4 * q;
        const bool rot = rope && 32 * cc < a.rope_cols;   // (rope_cols % 64 == 0: a chunk is rotated whole or not at all)
        const ff_f32x4 v = acc[ht];
        const float2 c2 = cs[2 * PAR + ht], s2 = sn[2 * PAR + ht];
        // products and sums rounded one by one, as the reference's x*c - y*s is (rope.go:81-105): contraction is off here
        const float y0 = v[0] * c2.x - v[1] * s2.x, y1 = v[0] * s2.x + v[1] * c2.x;
        const float y2 = v[2] * c2.y - v[3] * s2.y, y3 = v[2] * s2.y + v[3] * c2.y;
        const float4 o = make_float4(rot ? y0 : v[0], rot ? y1 : v[1], rot ? y2 : v[2], rot ? y3 : v[3]);
        *reinterpret_cast<float4*>(yrow + col) = o;   // (rows past M were clamped to row M-1 when loaded: they store its values again -- no store behind a branch inside the fenced steps)
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ff_f32x4 accA[2], accB[2];
    // iteration c (parity PAR): copy chunk c+2 into the slot chunk c-1 has left, multiply chunk c into `acc`, and in its shadow rotate + store chunk c-1 from `prev`
    auto iter = [&](int c, auto par, auto first, ff_f32x4 (&acc)[2], ff_f32x4 (&prev)[2]) {
        constexpr int PAR = decltype(par)::value;
        constexpr bool FIRST = decltype(first)::value != 0;
        if constexpr (!FIRST) {
            // every copy issued before the previous iteration has landed (this wave's: the wait leaves the previous iteration's 8 copies + 2 stores in
            // flight; the others': the barrier), and every wave has finished multiplying chunk c-1
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            __syncthreads();
        }
        const char* const src = img + (size_t)min(c + 2, nch - 1) * FF_W1B + wave * 1024 + lane * 16;
        char* const dst = lds + ((c + 2) % NS) * FF_W1B + wave * 1024;
        ff_gemm1(lds + (c % NS) * FF_W1B, r16, q, xh, xl, acc, [&](int s) {
            if ((s & 1) == 0) ff_dma1k(src + (s >> 1) * 4096, dst + (s >> 1) * 4096);   // (past the last chunk: the last one again, into a slot nobody reads any more)
            if constexpr (!FIRST) {
                if (s == 5) store_piece(prev, c - 1, std::integral_constant<int, 1 - PAR>{}, 0);
                if (s == 11) store_piece(prev, c - 1, std::integral_constant<int, 1 - PAR>{}, 1);
            }
        });
    };
    using ic0 = std::integral_constant<int, 0>;
    using ic1 = std::integral_constant<int, 1>;
    iter(0, ic0{}, ic1{}, accA, accB);
    iter(1, ic1{}, ic0{}, accB, accA);
    for (int c = 2; c < nch; c += 2) {
        iter(c, ic0{}, ic0{}, accA, accB);
        iter(c + 1, ic1{}, ic0{}, accB, accA);
    }
    store_piece(accB, nch - 1, std::integral_constant<int, 1>{}, 0);
    store_piece(accB, nch - 1, std::integral_constant<int, 1>{}, 1);
}

bool mimi_rowlin_supported(const RowLinArgs& a) {
    return a.img && a.K == FF_D && a.N >= 64 && a.N % 64 == 0 && a.M > 0 && aligned16(a.x) && a.xmap.ld % 4 == 0 && a.xmap.batch_stride % 4 == 0 && aligned16(a.y) &&
           a.ymap.ld % 4 == 0 && a.ymap.batch_stride % 4 == 0 && aligned16(a.img) && a.ln_w && a.ln_b && aligned16(a.ln_w) && aligned16(a.ln_b) &&
           (!a.rope_cos || (a.rope_sin && a.rope_cols % 64 == 0 && a.rope_cols <= a.N && (reinterpret_cast<uintptr_t>(a.rope_cos) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.rope_sin) & 7) == 0));
}

void launch_mimi_rowlin(const RowLinArgs& a, hipStream_t stream) {
    note_launch(a.rope_cos ? "k_mimi_rowlin+rope" : "k_mimi_rowlin");
    hipLaunchKernelGGL(k_mimi_rowlin, dim3((unsigned)((a.M + 63) / 64)), dim3(256), 0, stream, a);
}

bool mimi_ffn_supported(const FfnArgs& a) {
    return a.img && a.D == FF_D && a.F >= FF_CH && a.F % FF_CH == 0 && a.M > 0 && aligned16(a.x) && a.xmap.ld % 4 == 0 && a.xmap.batch_stride % 4 == 0 &&
           aligned16(a.img) && a.ln_w && a.ln_b && aligned16(a.ln_w) && aligned16(a.ln_b) && (!a.ls || aligned16(a.ls));
}

void launch_mimi_ffn(const FfnArgs& a, hipStream_t stream) {
    note_launch("k_mimi_ffn");
    const dim3 grid((unsigned)((a.M + 63) / 64));
    hipLaunchKernelGGL(k_mimi_ffn<0>, grid, dim3(256), 0, stream, a);   // (<1>: the no-copies ablation of tools/gpu_mimi_ab.sh's record, profiles/r4_mimi_ab.txt; instantiate it here to repeat it)
}

}  // namespace ptts
