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
// in this kernel uses it.
__device__ __forceinline__ void ff_dma32k(const char* src, char* dst, int wave, int lane) {
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)dst;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int idx = i * 4 + wave;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(base + idx * 1024), "v"(src + idx * 1024 + lane * 16) : "memory");
    }
}

}  // namespace

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
    {
        // k step s, lane (row, q): k = 32 s + 4 q + (0..3) and 32 s + 16 + 4 q + (0..3) -- the k order the W1 image is laid out in (and the order in
        // which a 16x16 accumulator tile pair would hand over its values: the same kernel can take its rows from a product later)
        float4 v[FF_KS][2];
#pragma unroll
        for (int s = 0; s < FF_KS; s++) {
            v[s][0] = *reinterpret_cast<const float4*>(xrow + 32 * s + 4 * q);
            v[s][1] = *reinterpret_cast<const float4*>(xrow + 32 * s + 16 + 4 * q);
        }
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < FF_KS; s++) sum += (v[s][0].x + v[s][0].y) + (v[s][0].z + v[s][0].w) + (v[s][1].x + v[s][1].y) + (v[s][1].z + v[s][1].w);
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / FF_D);
        float var = 0.f;
#pragma unroll
        for (int s = 0; s < FF_KS; s++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float d0 = v[s][h].x - mean, d1 = v[s][h].y - mean, d2 = v[s][h].z - mean, d3 = v[s][h].w - mean;
                var += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
        var += __shfl_xor(var, 16, 64);
        var += __shfl_xor(var, 32, 64);
        const float rstd = 1.0f / sqrtf(var * (1.0f / FF_D) + a.eps);
#pragma unroll
        for (int s = 0; s < FF_KS; s++) {
            float4 o[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int k = 32 * s + 16 * h + 4 * q;
                const float4 g = *reinterpret_cast<const float4*>(a.ln_w + k), b = *reinterpret_cast<const float4*>(a.ln_b + k);
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

    ff_f32x4 acc2[FF_OT];
#pragma unroll
    for (int t = 0; t < FF_OT; t++) acc2[t] = ff_f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses inside a slot.  W1 image: [k pair j (8)][hidden unit h (32)][128 B], the 16-byte chunk (k-step parity sg, lane group q) of
    // row h at position (4 sg + q) ^ ((h >> 1) & 7); W2 image: [output o (512)][64 B], lane group q's chunk at q ^ (3 * ((o >> 3) & 1)).
    const int sw1 = (r16 >> 1) & 7;
    const int w1_lane = r16 * 128;                                   // + ht * 2048 + (s >> 1) * 4096 + (((4 (s & 1) + q) ^ sw1) << 4)
    const int w2_lane = r16 * 64 + ((q ^ (((r16 >> 3) & 1) * 3)) << 4);   // + t * 1024

    // first product of one chunk: 16 k steps x (two fragment reads, four matrix instructions); the fragments of step s+1 are requested before step s is
    // multiplied.  piece(s) is vector work that rides in step s's shadow (the previous chunk's GELU, an eighth of it per two steps): each step is
    // fenced (sched_barrier) and its order pinned (sched_group_barrier) -- left to itself the scheduler issues read, wait, multiply back to back and
    // piles the vector work up in front of the matrix instructions instead of beside them.
    auto gemm1 = [&](const char* w1, ff_f32x4 (&acc)[2], auto&& piece) {
        acc[0] = ff_f32x4{0.f, 0.f, 0.f, 0.f};
        acc[1] = ff_f32x4{0.f, 0.f, 0.f, 0.f};
        FfFrag wa[2], wb[2];
        const char* wl = w1 + w1_lane;
        wa[0].q = *reinterpret_cast<const uint4*>(wl + ((q ^ sw1) << 4));
        wb[0].q = *reinterpret_cast<const uint4*>(wl + ((q ^ sw1) << 4) + 2048);
#pragma unroll
        for (int s = 0; s < FF_KS; s++) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < FF_KS) {
                const int off = ((s + 1) >> 1) * 4096 + (((4 * ((s + 1) & 1) + q) ^ sw1) << 4);
                wa[(s + 1) & 1].q = *reinterpret_cast<const uint4*>(wl + off);
                wb[(s + 1) & 1].q = *reinterpret_cast<const uint4*>(wl + off + 2048);
            }
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s & 1].v, xh[s].v, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[s & 1].v, xh[s].v, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s & 1].v, xl[s].v, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[s & 1].v, xl[s].v, acc[1], 0, 0, 0);
            piece(s);
            // (both accumulator chains are made to exist HERE: matrix instructions are pure values to the instruction selector, which otherwise lets the
            // second chain float to the end of the block -- behind the second product -- before the scheduler's fences ever see it)
            asm volatile("" : "+a"(acc[0]), "+a"(acc[1]));
            if (s + 1 < FF_KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the first copies has landed
    __syncthreads();                                   // ... and everybody else's
    ff_f32x4 acc1[2], acc1n[2];
    gemm1(w1s(0), acc1, [](int) {});

    for (int c = 0; c < nch; c++) {
        // every wave has finished iteration c-1 (its first product read W1 slot c&1 -- chunk c --, its second W2 slot (c-1)&1) and the copies issued
        // during it (W1 of chunk c+1, W2 of chunk c) have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 2 < nch) ff_dma32k(img + (size_t)(c + 2) * FF_IMG, w1s(c & 1), wave, lane);                  // W1 slot of chunk c is free
        if (c + 1 < nch) ff_dma32k(img + (size_t)(c + 1) * FF_IMG + FF_W1B, w2s((c + 1) & 1), wave, lane);   // W2 slot of chunk c-1 is free
        // The first product of the NEXT chunk and, in the shadow of its matrix instructions, this chunk's activation: GELU(erf) (tensor_util.go:84-94)
        // on the 8 sums the lane holds, split into hi + lo -> the second product's B operand.  (One basic block, no branch: the scheduler interleaves.)
        // this chunk's activation in the shadow of the next chunk's first product: GELU(erf) (tensor_util.go:84-94) on the 8 sums the lane holds (one per
        // two k steps), split into hi + lo (one pair per four steps) -> the second product's B operand
        FfFrag hh, hl;
        float g[8];
        gemm1(w1s(min(c + 1, nch - 1) & 1), acc1n, [&](int s) {   // (past the last chunk: the last slot once more, never used -- no branch around 64 matrix instructions)
            const int i = s >> 1;
            if ((s & 1) == 0) g[i] = ff_gelu(acc1[i >> 2][i & 3]);
            else if (i & 1) ff_split2(g[i - 1], g[i], hh.u[i >> 1], hl.u[i >> 1]);
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
            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t & 1].v, hl.v, acc2[t], 0, 0, 0);
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

bool mimi_ffn_supported(const FfnArgs& a) {
    return a.img && a.D == FF_D && a.F >= FF_CH && a.F % FF_CH == 0 && a.M > 0 && aligned16(a.x) && a.xmap.ld % 4 == 0 && a.xmap.batch_stride % 4 == 0 &&
           aligned16(a.img) && a.ln_w && a.ln_b && aligned16(a.ln_w) && aligned16(a.ln_b) && (!a.ls || aligned16(a.ls));
}

void launch_mimi_ffn(const FfnArgs& a, hipStream_t stream) {
    note_launch("k_mimi_ffn");
    hipLaunchKernelGGL(k_mimi_ffn, dim3((unsigned)((a.M + 63) / 64)), dim3(256), 0, stream, a);
}

}  // namespace ptts
