// gemv.hip -- the AR step's linear for a HANDFUL of rows (batch <= 4: BASELINE configs[1], the reference's own batch-1 case; linear.go:117-182,
// nn_ops.go:268-347 -- "one DotProduct per output").
//
// At one row the 16-row matrix tile of k_skinny is fifteen sixteenths padding and its block is a relay: rows -> LayerNorm -> bf16 hi / lo image in LDS
// -> barrier -> MFMA -> K parts summed through LDS -> barrier -> epilogue, every arrow a dependent step of a launch that moves only 32-128 KB per
// block.  Here a WAVE owns one output column: its 64 lanes hold the weight row (row-major, f32 or bf16, straight from HBM into registers: 16 bytes per
// lane and 256-deep step, one contiguous 1-KB burst per wave-instruction) and the activation row in the same layout, LayerNorm statistics are recomputed
// per wave (64 values per lane, one DPP reduction), the products are plain f32 FMAs -- exact f32 arithmetic like the reference's own dot product, no
// operand split -- and one DPP reduction ends in lane 0's epilogue and a 4-byte store.  No LDS, no barrier, nothing shared between waves: the launch is
// entry -> one round trip (weights, rows, epilogue operands all requested at once) -> FMAs -> store.
// Every prologue / epilogue form of the step is covered except the last launch (the Euler update with the step's bookkeeping and the next step's opening:
// 32 columns, stays on k_skinny); linear2's K = 4096 needs no split here, so at these batch sizes the step runs without split-K planes.
#include "kernels.h"
#include "device_util.h"

namespace ptts {

namespace {

constexpr int GV_MAXM = 4;

__device__ __forceinline__ float gv_dot4(float4 w, float4 x, float acc) {
    acc = fmaf(w.x, x.x, acc);
    acc = fmaf(w.y, x.y, acc);
    acc = fmaf(w.z, x.z, acc);
    return fmaf(w.w, x.w, acc);
}

}  // namespace

// NJ: 256-deep steps of K per lane (K <= 256 NJ); WBF16: weights bf16 (else f32); PRO: 0 none, 1 LayerNorm + affine, 2 LayerNorm + affine + adaLN modulation
template <bool WBF16, int PRO, int NJ>
__global__ __launch_bounds__(256) void k_gemv(GemmArgs a, SkinnyFuse fu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= a.N) return;   // (no barrier in this kernel)
    const bool tailcol = a.tail != nullptr && n == a.N - 1;   // the out_eos row stacked behind cond_embed: its weights live in their own row-major vector
    // ---- everything the wave needs is requested now: its weight row, the rows, the LayerNorm vectors, the epilogue's operands ----
    const char* wrow = tailcol ? (const char*)a.Wtail : (const char*)a.W + (int64_t)n * a.ldw * (WBF16 ? 2 : 4);
    float4 w[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int k = min(256 * j + 4 * lane, a.K - 4);   // (a step past K re-reads the last piece; its products are masked below)
        if constexpr (WBF16) {
            const uint2 u = *reinterpret_cast<const uint2*>(wrow + (int64_t)k * 2);
            w[j] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
        } else {
            w[j] = *reinterpret_cast<const float4*>(wrow + (int64_t)k * 4);
        }
        if (256 * j + 4 * lane >= a.K) w[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 lw[PRO ? NJ : 1], lb[PRO ? NJ : 1];
    if constexpr (PRO != 0) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int k = min(256 * j + 4 * lane, a.K - 4);
            lw[j] = fu.ln_w ? *reinterpret_cast<const float4*>(fu.ln_w + k) : make_float4(1.f, 1.f, 1.f, 1.f);
            lb[j] = fu.ln_b ? *reinterpret_cast<const float4*>(fu.ln_b + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const int nc = n;
    const float e_bias = a.bias ? a.bias[nc] : 0.f;
    const float e_addv = (a.addvec && !tailcol) ? a.addvec[nc] : 0.f;
    const float e_scl = a.scale ? a.scale[nc] : 1.f;
    float e_r[GV_MAXM], e_g[GV_MAXM];
#pragma unroll
    for (int m = 0; m < GV_MAXM; m++) {
        const int mm = min(m, a.M - 1);
        e_r[m] = (a.epi >= EPI_RESADD && !tailcol) ? a.R[(int64_t)mm * a.cmap.ld + nc] : 0.f;
        e_g[m] = (a.epi == EPI_GATE_RESADD && !tailcol) ? a.gate[(int64_t)mm * a.ldg + nc] : 0.f;
    }
    const float rk = 1.0f / (float)a.K;
#pragma unroll
    for (int m = 0; m < GV_MAXM; m++) {
        if (m >= a.M) break;
        const float* xrow = a.A + (int64_t)m * a.amap.ld;
        float4 x[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int k = min(256 * j + 4 * lane, a.K - 4);
            x[j] = *reinterpret_cast<const float4*>(xrow + k);
            if (256 * j + 4 * lane >= a.K) x[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (PRO != 0) {   // LayerNorm over the row (biased variance, linear.go:295-309), optionally adaLN-modulated (tensor_util.go:175-193)
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++) s += (x[j].x + x[j].y) + (x[j].z + x[j].w);
            const float mean = wave_sum_dpp(s) * rk;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                if (256 * j + 4 * lane < a.K) {
                    const float d0 = x[j].x - mean, d1 = x[j].y - mean, d2 = x[j].z - mean, d3 = x[j].w - mean;
                    v += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            }
            const float rstd = 1.0f / sqrtf(wave_sum_dpp(v) * rk + fu.eps);
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int k = min(256 * j + 4 * lane, a.K - 4);
                float4 o;
                o.x = (x[j].x - mean) * rstd * lw[j].x + lb[j].x;
                o.y = (x[j].y - mean) * rstd * lw[j].y + lb[j].y;
                o.z = (x[j].z - mean) * rstd * lw[j].z + lb[j].z;
                o.w = (x[j].w - mean) * rstd * lw[j].w + lb[j].w;
                if constexpr (PRO == 2) {
                    const float4 sc = *reinterpret_cast<const float4*>(fu.scale + (int64_t)m * fu.ldmod + k);
                    const float4 sh = *reinterpret_cast<const float4*>(fu.shift + (int64_t)m * fu.ldmod + k);
                    o.x = o.x * (1.0f + sc.x) + sh.x; o.y = o.y * (1.0f + sc.y) + sh.y; o.z = o.z * (1.0f + sc.z) + sh.z; o.w = o.w * (1.0f + sc.w) + sh.w;
                }
                if (256 * j + 4 * lane >= a.K) o = make_float4(0.f, 0.f, 0.f, 0.f);
                x[j] = o;
                if (fu.y_out && n == 0 && 256 * j + 4 * lane < a.K) *reinterpret_cast<float4*>(fu.y_out + (int64_t)m * a.K + k) = o;   // the normalised rows, once
            }
        }
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; j++) acc = gv_dot4(w[j], x[j], acc);
        acc = wave_sum_dpp(acc);
        if (lane == 0) {
            float v = acc + e_bias;
            if (tailcol) { a.tail[m] = v; continue; }
            switch (a.epi) {
                case EPI_NONE: break;
                case EPI_GELU: v = gelu1(v); break;
                case EPI_SILU: v = silu1(e_addv + v); break;
                case EPI_ELU: v = elu1(v); break;
                case EPI_RESADD: v = e_r[m] + v; break;
                case EPI_SCALE_RESADD: v = e_r[m] + e_scl * v; break;
                case EPI_GATE_RESADD: v = e_r[m] + e_g[m] * v; break;
                case EPI_AXPY: v = e_r[m] + a.alpha * v; break;
                case EPI_RESADD_ELU: v = elu1(e_r[m] + v); break;
            }
            a.C[(int64_t)m * a.cmap.ld + n] = v;
        }
    }
}

// rows in flat dense layouts, a row-major weight matrix (and, with `tail`, the stacked last row as its own vector), no pending split-K planes, no bookkeeping
bool gemv_supported(const GemmArgs& a, const SkinnyFuse& fu) {
    const bool ln = fu.ln != 0;
    return a.M >= 1 && a.M <= GV_MAXM && a.W && !a.wt_i8 && a.K % 4 == 0 && a.K >= 4 && a.K <= 4096 && a.aop == AOP_NONE && a.amap.rows_per_batch == 0 &&
           a.cmap.rows_per_batch == 0 && a.amap.ld % 4 == 0 && a.ldw % 4 == 0 && aligned16(a.A) && aligned16(a.W) && !a.kslice && !a.rope_cos &&
           (!a.tail || (a.Wtail && ((reinterpret_cast<uintptr_t>(a.Wtail) & 15) == 0))) && !fu.partial && !fu.fin && !fu.x_out &&
           (!ln || (a.amap.ld == a.K && (!fu.ln_w == !fu.ln_b) && (!fu.scale || (fu.shift && fu.ldmod % 4 == 0)) && (!fu.ln_w || aligned16(fu.ln_w)))) &&
           (ln || (!fu.scale && !fu.ln_w));
}

template <bool WBF16, int PRO>
static void launch_gemv_nj(const GemmArgs& a, const SkinnyFuse& fu, hipStream_t stream) {
    const dim3 grid((unsigned)((a.N + 3) / 4));
    const int nj = (a.K + 255) / 256;
    if (nj <= 2) hipLaunchKernelGGL((k_gemv<WBF16, PRO, 2>), grid, dim3(256), 0, stream, a, fu);
    else if (nj <= 4) hipLaunchKernelGGL((k_gemv<WBF16, PRO, 4>), grid, dim3(256), 0, stream, a, fu);
    else if (nj <= 8) hipLaunchKernelGGL((k_gemv<WBF16, PRO, 8>), grid, dim3(256), 0, stream, a, fu);
    else hipLaunchKernelGGL((k_gemv<WBF16, PRO, 16>), grid, dim3(256), 0, stream, a, fu);
}

void launch_gemv(const GemmArgs& a, const SkinnyFuse& fu, hipStream_t stream) {
    note_launch("k_gemv");
    const int pro = !fu.ln ? 0 : (fu.scale ? 2 : 1);
    if (a.w_bf16) {
        if (pro == 0) launch_gemv_nj<true, 0>(a, fu, stream);
        else if (pro == 1) launch_gemv_nj<true, 1>(a, fu, stream);
        else launch_gemv_nj<true, 2>(a, fu, stream);
    } else {
        if (pro == 0) launch_gemv_nj<false, 0>(a, fu, stream);
        else if (pro == 1) launch_gemv_nj<false, 1>(a, fu, stream);
        else launch_gemv_nj<false, 2>(a, fu, stream);
    }
}

}  // namespace ptts
