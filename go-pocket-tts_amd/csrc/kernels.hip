// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the PocketTTS synthesis path.
//
// Each kernel names the reference computation it replaces.
// Layout conventions: activations are row-major [rows, channels] ("channels-last" for the
// Mimi decoder, so that a causal convolution window is one contiguous span and every
// global access is a coalesced 16-byte-per-lane stream); weights are [out, in] row-major.
#include <cstdlib>
#include "kernels.h"
#include "device_util.h"
#include "step_open.h"

namespace ptts {

// ------------------------------------------------------------------------------------------------
// GEMM  C[M,N] = epi( aop(A)[M,K] * W[N,K]^T )       (K4, K8, K11, K13-K17: every Linear / Conv1d /
// ConvTranspose1d of the reference: linear.go:117-182, conv1d.go:20-83, convtranspose1d.go:73-148)
//
// 64x64 block tile, 4 waves as 2x2, each wave one 32x32 accumulator of v_mfma_f32_32x32x2_f32
// (exact f32 FMA chain -- the arithmetic the reference's f32 dot products perform, in a different
// summation order).  K is walked in 32-deep tiles staged through LDS with 16-byte accesses; the
// k index inside a tile is permuted consistently for both operands (lane half kq owns
// k = 8*g + 4*kq + j), which lets each lane fetch its four values per operand with one ds_read_b128.
// ------------------------------------------------------------------------------------------------
constexpr int GM = 64, GN = 64, GK = 32, GLD = GK + 4;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool WBF16>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs a, int a_vec, int w_vec) {
    __shared__ __attribute__((aligned(16))) float As[GM * GLD];
    __shared__ __attribute__((aligned(16))) float Ws[GN * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
    const int wm = wave >> 1, wn = wave & 1;

    // staging assignment: 2 x (row, c4) per thread for each operand
    int srow[2], sc4[2];
    const float* aptr[2];
    const char* wptr[2];
    bool arow_ok[2], wrow_ok[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        int idx = tid + i * 256;
        srow[i] = idx >> 3;
        sc4[i] = (idx & 7) * 4;
        int gm = m0 + srow[i], gn = n0 + srow[i];
        arow_ok[i] = gm < a.M;
        wrow_ok[i] = gn < a.N;
        aptr[i] = a.A + (arow_ok[i] ? row_off(a.amap, gm) : 0);
        wptr[i] = (const char*)a.W + (wrow_ok[i] ? (int64_t)gn * a.ldw * (WBF16 ? 2 : 4) : 0);
    }

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;

    for (int k0 = 0; k0 < a.K; k0 += GK) {
        float4 av[2], wv[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            int k = k0 + sc4[i];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (arow_ok[i]) {
                if (a_vec && k + 3 < a.K) v = *reinterpret_cast<const float4*>(aptr[i] + k);
                else {
                    if (k + 0 < a.K) v.x = aptr[i][k + 0];
                    if (k + 1 < a.K) v.y = aptr[i][k + 1];
                    if (k + 2 < a.K) v.z = aptr[i][k + 2];
                    if (k + 3 < a.K) v.w = aptr[i][k + 3];
                }
                if (a.aop == AOP_ELU) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
            }
            av[i] = v;
            float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
            if (wrow_ok[i]) {
                if (WBF16) {
                    const unsigned short* wp = reinterpret_cast<const unsigned short*>(wptr[i]);
                    if (w_vec && k + 3 < a.K) {
                        uint2 raw = *reinterpret_cast<const uint2*>(wp + k);
                        u.x = __uint_as_float(raw.x << 16); u.y = __uint_as_float(raw.x & 0xffff0000u);
                        u.z = __uint_as_float(raw.y << 16); u.w = __uint_as_float(raw.y & 0xffff0000u);
                    } else {
                        if (k + 0 < a.K) u.x = bf16_bits_to_f32(wp[k + 0]);
                        if (k + 1 < a.K) u.y = bf16_bits_to_f32(wp[k + 1]);
                        if (k + 2 < a.K) u.z = bf16_bits_to_f32(wp[k + 2]);
                        if (k + 3 < a.K) u.w = bf16_bits_to_f32(wp[k + 3]);
                    }
                } else {
                    const float* wp = reinterpret_cast<const float*>(wptr[i]);
                    if (w_vec && k + 3 < a.K) u = *reinterpret_cast<const float4*>(wp + k);
                    else {
                        if (k + 0 < a.K) u.x = wp[k + 0];
                        if (k + 1 < a.K) u.y = wp[k + 1];
                        if (k + 2 < a.K) u.z = wp[k + 2];
                        if (k + 3 < a.K) u.w = wp[k + 3];
                    }
                }
            }
            wv[i] = u;
        }
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 2; i++) {
            *reinterpret_cast<float4*>(&As[srow[i] * GLD + sc4[i]]) = av[i];
            *reinterpret_cast<float4*>(&Ws[srow[i] * GLD + sc4[i]]) = wv[i];
        }
        __syncthreads();
        const int r = lane & 31, kq = lane >> 5;
#pragma unroll
        for (int g = 0; g < GK / 8; g++) {
            float4 x4 = *reinterpret_cast<const float4*>(&As[(wm * 32 + r) * GLD + g * 8 + kq * 4]);
            float4 w4 = *reinterpret_cast<const float4*>(&Ws[(wn * 32 + r) * GLD + g * 8 + kq * 4]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4.x, w4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4.y, w4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4.z, w4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4.w, w4.w, acc, 0, 0, 0);
        }
    }

    // epilogue: lane holds C[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31] of its 32x32 tile
    const int n = n0 + wn * 32 + (lane & 31);
    if (n >= a.N) return;
    const float bias = a.bias ? a.bias[n] : 0.0f;
    const float addv = a.addvec ? a.addvec[n] : 0.0f;
    const float scl = a.scale ? a.scale[n] : 1.0f;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        int m = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (m >= a.M) continue;
        float v = acc[reg] + bias;
        int64_t co = row_off(a.cmap, m) + n;
        switch (a.epi) {
            case EPI_NONE: break;
            case EPI_GELU: v = gelu1(v); break;
            case EPI_SILU: v = silu1(addv + v); break;
            case EPI_ELU: v = elu1(v); break;
            case EPI_RESADD: v = a.R[co] + v; break;
            case EPI_SCALE_RESADD: v = a.R[co] + scl * v; break;
            case EPI_GATE_RESADD: v = a.R[co] + a.gate[(int64_t)m * a.ldg + n] * v; break;
            case EPI_AXPY: v = a.R[co] + a.alpha * v; break;
            case EPI_RESADD_ELU: v = elu1(a.R[co] + v); break;
        }
        a.C[co] = v;
    }
}

bool launch_gemm_rope(const GemmArgs& a, hipStream_t stream) {
    if (gemm5_supported(a)) { launch_gemm5(a, stream); return true; }
    if (!gemm3_supported(a)) return false;
    launch_gemm3(a, stream);
    return true;
}

void launch_gemm(const GemmArgs& a, hipStream_t stream) {
    if (a.M <= 0 || a.N <= 0) return;
    // a few rows (prefill of a short prompt, batch-1 serving): the step's weight-streaming kernel beats a tile GEMM that would
    // fill a handful of CUs
    if (a.M <= 64 && a.Wt && skinny_supported(a, 1)) { launch_skinny(a, SkinnyFuse{}, 1, nullptr, stream); return; }
    // a few dozen to a few hundred rows (the prompt of a handful of newcomers joining a running batch): the same kernel over 64-row chunks -- each chunk
    // streams the weights again (a few MB: 6-9 us), which the 128-row tile GEMMs below cannot match at these sizes (k_gemm2 / k_gemm: 90-120 us per
    // launch at 75 rows in the serving trace of round 4, 17 % of the GPU's time while serving)
    if (a.M <= kSkinnyChunkRows && a.Wt && !a.rope_cos && !a.kslice && !a.tail && a.amap.rows_per_batch == 0 && a.cmap.rows_per_batch == 0) {
        GemmArgs c = a;
        c.M = 64;
        if (skinny_supported(c, 1)) {
            for (int m0 = 0; m0 < a.M; m0 += 64) {
                c = a;
                c.M = std::min(64, a.M - m0);
                c.A = a.A + (int64_t)m0 * a.amap.ld;
                c.C = a.C + (int64_t)m0 * a.cmap.ld;
                if (a.R) c.R = a.R + (int64_t)m0 * a.cmap.ld;
                if (a.gate) c.gate = a.gate + (int64_t)m0 * a.ldg;
                launch_skinny(c, SkinnyFuse{}, 1, nullptr, stream);
            }
            return;
        }
    }
    if (gemm_wres_supported(a)) { launch_gemm_wres(a, stream); return; }
    if (gemm5_supported(a)) { launch_gemm5(a, stream); return; }
    if (gemm3_supported(a)) { launch_gemm3(a, stream); return; }
    if (gemm2_supported(a)) { launch_gemm2(a, stream); return; }
    note_launch("k_gemm");
    int a_vec = aligned16(a.A) && a.amap.ld % 4 == 0 && a.amap.batch_stride % 4 == 0;
    int w_vec = a.w_bf16 ? ((reinterpret_cast<uintptr_t>(a.W) & 7) == 0 && a.ldw % 4 == 0) : (aligned16(a.W) && a.ldw % 4 == 0);
    dim3 grid((a.N + GN - 1) / GN, (a.M + GM - 1) / GM);
    if (a.w_bf16) hipLaunchKernelGGL(k_gemm<true>, grid, dim3(256), 0, stream, a, a_vec, w_vec);
    else hipLaunchKernelGGL(k_gemm<false>, grid, dim3(256), 0, stream, a, a_vec, w_vec);
}

// ------------------------------------------------------------------------------------------------
// small elementwise / gather kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_embed_gather(const float* table, const int64_t* ids, int n, int d, float* out) {  // K1
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t tot = (int64_t)n * (d / 4);
    if (i >= tot) return;
    int r = (int)(i / (d / 4)), c = (int)(i % (d / 4));
    reinterpret_cast<float4*>(out + (int64_t)r * d)[c] = reinterpret_cast<const float4*>(table + ids[r] * (int64_t)d)[c];
}
void launch_embed_gather(const float* table, const int64_t* ids, int n, int d, float* out, hipStream_t stream) {
    if (n <= 0) return;
    int64_t tot = (int64_t)n * (d / 4);
    hipLaunchKernelGGL(k_embed_gather, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, table, ids, n, d, out);
}

__global__ void k_replace_nan(const float* in, const float* bos, int rows, int d, float* out) {  // K2 (first half)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * d) return;
    float v = in[i];
    out[i] = isnan(v) ? bos[i % d] : v;
}
void launch_replace_nan(const float* in, const float* bos, int rows, int d, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(k_replace_nan, dim3((rows * d + 255) / 256), dim3(256), 0, stream, in, bos, rows, d, out);
}

__global__ void k_silu(float* x, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = silu1(x[i]);
}
void launch_silu(float* x, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_silu, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, n);
}

__global__ void k_copy_rows(const float* src, int64_t lds, float* dst, int64_t ldd, int rows, int d) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * d) return;
    int r = (int)(i / d), c = (int)(i % d);
    dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds + c];
}
void launch_copy_rows(const float* src, int64_t lds, float* dst, int64_t ldd, int rows, int d, hipStream_t stream) {
    int64_t n = (int64_t)rows * d;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, lds, dst, ldd, rows, d);
}

__global__ void k_timestep_features(float t, const float* freqs, int nf, float* out) {  // flow_net.go:52-65
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    float arg = t * freqs[i];
    out[i] = (float)cos((double)arg);       // mimi.go:796-797: the reference evaluates cos/sin in f64
    out[nf + i] = (float)sin((double)arg);
}
void launch_timestep_features(float t, const float* freqs, int nf, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(k_timestep_features, dim3((nf + 63) / 64), dim3(64), 0, stream, t, freqs, nf, out);
}

__global__ void k_avg2(const float* a, const float* b, float* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (a[i] + b[i]) * 0.5f;
}
void launch_avg2(const float* a, const float* b, float* y, int n, hipStream_t stream) {
    hipLaunchKernelGGL(k_avg2, dim3((n + 255) / 256), dim3(256), 0, stream, a, b, y, n);
}

// The continuous engine's decode input: the frames of up to 128 finished utterances from their staging rows into one packed [utterance][T][ld] block, the padding
// behind an utterance's last frame zeroed -- one launch with its table in the kernel arguments (as a memset + one hipMemcpyAsync per utterance the step stream
// stood still for ~2.5 ms per decode: 25 copy-engine round trips of ~100 us each, profiles/r5_serve_sweep.txt "group traces").
__global__ __launch_bounds__(256) void k_gather_frames(GatherTable t, const float* stage, int64_t row_stride, float* dst, int ld) {
    const GatherTable::Row r = t.rows[blockIdx.x];
    const float4* src = reinterpret_cast<const float4*>(stage + (int64_t)r.src_row * row_stride);
    float4* out = reinterpret_cast<float4*>(dst + r.dst_off);
    const int n_copy = r.nf * ld / 4, n_all = r.T * ld / 4;
    for (int i = threadIdx.x; i < n_all; i += 256) out[i] = i < n_copy ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
}
void launch_gather_frames(const GatherTable& t, int n, const float* stage, int64_t row_stride, float* dst, int ld, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_frames, dim3((unsigned)n), dim3(256), 0, stream, t, stage, row_stride, dst, ld);
}

// The continuous engine's read-back of the slots' counters: a kernel that stores them straight into page-locked host memory, in order on the step stream.  (As
// hipMemcpyAsync the read-back went through the copy engine, behind whatever PCM the decoder's stream had queued there: the step stream stood still for up to
// 2.5 ms in front of every group that followed a decode start, PTTS_CONT_TRACE.)
__global__ void k_readback_i32(const int32_t* a, int na, const int32_t* b, int nb, int32_t* host) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) host[i] = a[i];
    else if (i < na + nb) host[i] = b[i - na];
}
void launch_readback_i32(const int32_t* a, int na, const int32_t* b, int nb, int32_t* host, hipStream_t stream) {
    const int n = na + (b ? nb : 0);
    if (n <= 0) return;
    hipLaunchKernelGGL(k_readback_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, na, b, b ? nb : 0, host);
}

__global__ void k_fill_i32(int32_t* p, int32_t v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
void launch_fill_i32(int32_t* p, int32_t v, int n, hipStream_t stream) {
    hipLaunchKernelGGL(k_fill_i32, dim3((n + 255) / 256), dim3(256), 0, stream, p, v, n);
}
__global__ void k_add_i32(int32_t* p, const int32_t* inc, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += inc[i];
}
void launch_add_i32(int32_t* p, const int32_t* inc, int n, hipStream_t stream) {
    hipLaunchKernelGGL(k_add_i32, dim3((n + 255) / 256), dim3(256), 0, stream, p, inc, n);
}

// Continuous batching (continuous.cpp): a slot of a running batch is handed to a new utterance / taken from one that was cancelled.
// One thread per slot touched; the live-utterance counter moves by the number of slots whose `active` flag changes.
__global__ void k_slot_admit(StepState s, int32_t* pre_len, const void** pre_k, const void** pre_v, const SlotAdmit* a, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int sl = a[i].slot;
    s.kv_len[sl] = a[i].kv_len; pre_len[sl] = a[i].pre_len; pre_k[sl] = a[i].pre_k; pre_v[sl] = a[i].pre_v;
    s.active[sl] = 1; s.step[sl] = 0; s.countdown[sl] = -1; s.n_frames[sl] = 0; s.eos_step[sl] = -1; s.broke[sl] = 0;
    s.max_steps[sl] = a[i].max_steps; s.frames_after_eos[sl] = a[i].frames_after_eos; s.eos_threshold[sl] = a[i].eos_threshold;
    if (i == 0) atomicAdd(s.n_active, n);
}
void launch_slot_admit(const StepState& s, int32_t* pre_len, const void** pre_k, const void** pre_v, const SlotAdmit* dev, int n, hipStream_t stream) {
    if (n > 0) hipLaunchKernelGGL(k_slot_admit, dim3((n + 63) / 64), dim3(64), 0, stream, s, pre_len, pre_k, pre_v, dev, n);
}
__global__ void k_slot_retire(StepState s, const int32_t* slots, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int sl = slots[i];
    if (s.active[sl]) { s.active[sl] = 0; atomicSub(s.n_active, 1); }
}
void launch_slot_retire(const StepState& s, const int32_t* slots_dev, int n, hipStream_t stream) {
    if (n > 0) hipLaunchKernelGGL(k_slot_retire, dim3((n + 63) / 64), dim3(64), 0, stream, s, slots_dev, n);
}

// ------------------------------------------------------------------------------------------------
// RoPE on rows of a qkv buffer (K5; rope.go:81-105): interleaved pairs, table row = position
// ------------------------------------------------------------------------------------------------
__global__ void k_rope_rows(float* x, RowMap xmap, int col0, int heads, int hd, const int32_t* pos, int pos_base,
                            int rows_per_seg, int rows, const float* cos_t, const float* sin_t) {
    const int half = hd / 2;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t tot = (int64_t)rows * heads * half;
    if (i >= tot) return;
    int j = (int)(i % half);
    int h = (int)((i / half) % heads);
    int r = (int)(i / ((int64_t)half * heads));
    int p = pos ? pos[r] : pos_base + (rows_per_seg ? r % rows_per_seg : r);
    float* v = x + row_off(xmap, r) + col0 + h * hd + 2 * j;
    float a = v[0], b = v[1];
    float c = cos_t[(int64_t)p * half + j], s = sin_t[(int64_t)p * half + j];
    {
#pragma clang fp contract(off)
        v[0] = a * c - b * s;   // rounded product by product (the GEMM epilogues' form: gemm3.hip / gemm5.hip)
        v[1] = a * s + b * c;
    }
}
void launch_rope_rows(float* x, RowMap xmap, int col0, int heads, int hd, const int32_t* pos, int pos_base, int rows_per_seg,
                      int rows, const float* cos_t, const float* sin_t, hipStream_t stream) {
    int64_t tot = (int64_t)rows * heads * (hd / 2);
    if (tot <= 0) return;
    hipLaunchKernelGGL(k_rope_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, x, xmap, col0, heads, hd, pos,
                       pos_base, rows_per_seg, rows, cos_t, sin_t);
}

// K6: append (already rotated) K and V rows into the cache [slot][head][cap][hd]  (flow_transformer.go:32-67)
template <bool KVBF16>
__global__ void k_kv_append(const float* qkv, int64_t ld, int d_model, int heads, int hd, const int32_t* row_slot,
                            const int32_t* row_pos, int rows, void* kc, void* vc, int64_t cap) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t tot = (int64_t)rows * d_model;
    if (i >= tot) return;
    int c = (int)(i % d_model), r = (int)(i / d_model);
    int h = c / hd, e = c % hd;
    int64_t dst = (((int64_t)row_slot[r] * heads + h) * cap + row_pos[r]) * hd + e;
    float kv = qkv[(int64_t)r * ld + d_model + c], vv = qkv[(int64_t)r * ld + 2 * d_model + c];
    if (KVBF16) {
        reinterpret_cast<unsigned short*>(kc)[dst] = f32_to_bf16_bits(kv);
        reinterpret_cast<unsigned short*>(vc)[dst] = f32_to_bf16_bits(vv);
    } else {
        reinterpret_cast<float*>(kc)[dst] = kv;
        reinterpret_cast<float*>(vc)[dst] = vv;
    }
}
void launch_kv_append(const float* qkv, int64_t ld, int d_model, int heads, int hd, const int32_t* row_slot,
                      const int32_t* row_pos, int rows, void* kcache, void* vcache, int kv_bf16, int64_t cap,
                      hipStream_t stream) {
    int64_t tot = (int64_t)rows * d_model;
    if (tot <= 0) return;
    dim3 g((unsigned)((tot + 255) / 256));
    if (kv_bf16) hipLaunchKernelGGL(k_kv_append<true>, g, dim3(256), 0, stream, qkv, ld, d_model, heads, hd, row_slot, row_pos, rows, kcache, vcache, cap);
    else hipLaunchKernelGGL(k_kv_append<false>, g, dim3(256), 0, stream, qkv, ld, d_model, heads, hd, row_slot, row_pos, rows, kcache, vcache, cap);
}

// ------------------------------------------------------------------------------------------------
// Attention with absolute positions (K7; attention.go:307-484), head_dim 64.
//   key j is visible to a query at position p iff 0 <= p - j (and p - j < context when context >= 0);
//   cache slots beyond p are never read (the reference masks them with posK = -1 before the dot
//   product, so NaN padding in voice-state caches cannot propagate); an empty key set gives zeros.
// One query per (row, head).  WPQ waves cooperate on a query (4 for the latency-bound AR step,
// 1 for prefill / Mimi where there are many queries).  Inside a wave 16 lanes share one key
// (16 B each = one coalesced 256-byte f32 row or 128-byte bf16 row), 4 keys per wave-instruction.
// Scores go through LDS once; softmax max/sum are wave reductions; P*V partials are combined in a
// fixed order, so results are bitwise reproducible.
// With fused_step the block first rotates q/k of its (slot, head) by the table row at the cache
// offset and appends k, v at that offset (flow_transformer.go:340-347) -- K5 + K6 + K7 in one launch.
// ------------------------------------------------------------------------------------------------
template <bool KVBF16> __device__ __forceinline__ float4 load_kv4(const void* base, int64_t elem_off) {
    if (KVBF16) {
        uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem_off);
        return make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                           __uint_as_float(raw.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem_off);
}

template <int WPQ, bool KVBF16>
__global__ __launch_bounds__(256) void k_attention(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int QPB = 4 / WPQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = wave / WPQ, wq = wave % WPQ;          // query within block, wave within query
    const int h = blockIdx.x;
    const int row = blockIdx.y * QPB + qi;
    const bool row_ok = row < a.rows;
    float* sc = smem + (size_t)qi * a.max_keys;                          // scores of this query
    float* part = smem + (size_t)QPB * a.max_keys + (size_t)qi * WPQ * 64;  // [WPQ][64] partial outputs
    float* qs = smem + (size_t)QPB * a.max_keys + (size_t)QPB * WPQ * 64;   // [64] rotated q (fused step)

    int seg = 0, pos = -1;
    if (row_ok) {
        seg = a.row_seg ? a.row_seg[row] : (a.rows_per_seg ? row / a.rows_per_seg : row);
        pos = a.seg_len ? a.seg_len[seg] : (a.row_pos ? a.row_pos[row] : a.pos_base + (a.rows_per_seg ? row % a.rows_per_seg : row));
    }
    const bool live = row_ok && (!a.active || a.active[seg]);
    const char* kbase = (const char*)a.k + ((int64_t)seg * a.k_seg_stride + (int64_t)h * a.k_head_stride) * (KVBF16 ? 2 : 4);
    const char* vbase = (const char*)a.v + ((int64_t)seg * a.k_seg_stride + (int64_t)h * a.k_head_stride) * (KVBF16 ? 2 : 4);

    if (a.fused_step) {  // WPQ == 4, one query per block: uniform control flow
        if (live) {
            const float* qr = a.qkv + (int64_t)row * a.qkv_ld + h * 64;
            if (tid < 32) {
                float c = a.cos_t[(int64_t)pos * 32 + tid], s = a.sin_t[(int64_t)pos * 32 + tid];
                float q0 = qr[2 * tid], q1 = qr[2 * tid + 1];
                qs[2 * tid] = q0 * c - q1 * s;
                qs[2 * tid + 1] = q0 * s + q1 * c;
                float k0 = qr[a.d_model + 2 * tid], k1 = qr[a.d_model + 2 * tid + 1];
                float r0 = k0 * c - k1 * s, r1 = k0 * s + k1 * c;
                int64_t dst = (int64_t)pos * 64 + 2 * tid;
                if (KVBF16) {
                    reinterpret_cast<unsigned short*>(const_cast<char*>(kbase))[dst] = f32_to_bf16_bits(r0);
                    reinterpret_cast<unsigned short*>(const_cast<char*>(kbase))[dst + 1] = f32_to_bf16_bits(r1);
                } else {
                    reinterpret_cast<float*>(const_cast<char*>(kbase))[dst] = r0;
                    reinterpret_cast<float*>(const_cast<char*>(kbase))[dst + 1] = r1;
                }
            } else if (tid < 96) {
                int e = tid - 32;
                float vv = qr[2 * a.d_model + e];
                int64_t dst = (int64_t)pos * 64 + e;
                if (KVBF16) reinterpret_cast<unsigned short*>(const_cast<char*>(vbase))[dst] = f32_to_bf16_bits(vv);
                else reinterpret_cast<float*>(const_cast<char*>(vbase))[dst] = vv;
            }
        }
        __syncthreads();
    }

    int j0 = 0, nk = 0;
    if (live) {
        j0 = a.context >= 0 ? max(0, pos - a.context + 1) : 0;
        nk = pos - j0 + 1;
        if (nk < 0) nk = 0;
    }
    const int sub = lane & 15, kq = lane >> 4;
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        if (a.fused_step) q4 = *reinterpret_cast<const float4*>(qs + sub * 4);
        else q4 = *reinterpret_cast<const float4*>(a.q + row_off(RowMap{a.q_ld, a.q_rows_per_batch, a.q_batch_stride}, row) + a.q_col0 + h * 64 + sub * 4);
    }
    const float scale = 0.125f;  // 1/sqrt(64)

    // pass 1: scores.  16 keys per wave-iteration: the four 1-KiB key loads are issued before any is consumed
    for (int base = wq * 16; base < nk; base += WPQ * 16) {
        float4 k4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int jj = base + u * 4 + kq;
            k4[u] = jj < nk ? load_kv4<KVBF16>(kbase, (int64_t)(j0 + jj) * a.k_row_stride + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int jj = base + u * 4 + kq;
            float p = q4.x * k4[u].x + q4.y * k4[u].y + q4.z * k4[u].z + q4.w * k4[u].w;
            p += __shfl_xor(p, 8, WAVE);
            p += __shfl_xor(p, 4, WAVE);
            p += __shfl_xor(p, 2, WAVE);
            p += __shfl_xor(p, 1, WAVE);
            if (sub == 0 && jj < nk) sc[jj] = p * scale;
        }
    }
    __syncthreads();
    // softmax statistics (every wave of the query computes the same values)
    float mx = -INFINITY;
    for (int j = lane; j < nk; j += WAVE) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int j = lane; j < nk; j += WAVE) sum += expf(sc[j] - mx);
    sum = wave_sum(sum);
    // pass 2: P * V, same 16-keys-in-flight shape
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = wq * 16; base < nk; base += WPQ * 16) {
        float4 v4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int jj = base + u * 4 + kq;
            v4[u] = jj < nk ? load_kv4<KVBF16>(vbase, (int64_t)(j0 + jj) * a.k_row_stride + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int jj = base + u * 4 + kq;
            if (jj < nk) {
                float p = expf(sc[jj] - mx);
                o.x += p * v4[u].x; o.y += p * v4[u].y; o.z += p * v4[u].z; o.w += p * v4[u].w;
            }
        }
    }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        o.x += __shfl_xor(o.x, off, WAVE); o.y += __shfl_xor(o.y, off, WAVE);
        o.z += __shfl_xor(o.z, off, WAVE); o.w += __shfl_xor(o.w, off, WAVE);
    }
    if (WPQ > 1) {
        if (kq == 0) *reinterpret_cast<float4*>(part + wq * 64 + sub * 4) = o;
        __syncthreads();
        if (wq == 0 && kq == 0) {
            o = *reinterpret_cast<const float4*>(part + sub * 4);
#pragma unroll
            for (int w = 1; w < WPQ; w++) {
                float4 t = *reinterpret_cast<const float4*>(part + w * 64 + sub * 4);
                o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
            }
        }
    }
    if (row_ok && wq == 0 && kq == 0) {
        float inv = (nk > 0 && sum > 0.0f) ? 1.0f / sum : 0.0f;
        float4 r = nk > 0 ? make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(a.out + row_off(RowMap{a.out_ld, a.o_rows_per_batch, a.o_batch_stride}, row) + h * 64 + sub * 4) = r;
    }
}

thread_local std::map<std::string, int64_t>* g_launch_census = nullptr;
thread_local const char* g_last_attn_kernel = "";   // which kernel the calling thread's last launch_attention picked (parity tests assert it)

void launch_attention(const AttnArgs& a, hipStream_t stream) {
    if (a.rows <= 0) return;
    if (attn_step_supported(a)) { g_last_attn_kernel = "k_attn_step"; launch_attn_step(a, stream); return; }
    if (attn_window_supported(a)) { g_last_attn_kernel = a.rag_off ? "k_attn_window<ragged>" : "k_attn_window"; launch_attn_window(a, stream); return; }
    g_last_attn_kernel = "k_attention";
    note_launch("k_attention");
    const int wpq = a.fused_step ? 4 : (a.rows * a.heads < 2048 ? 4 : 1);
    const int qpb = 4 / wpq;
    size_t lds = ((size_t)qpb * a.max_keys + (size_t)qpb * wpq * 64 + 64) * sizeof(float);
    dim3 grid(a.heads, (a.rows + qpb - 1) / qpb);
    if (wpq == 4) {
        if (a.kv_bf16) hipLaunchKernelGGL((k_attention<4, true>), grid, dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((k_attention<4, false>), grid, dim3(256), lds, stream, a);
    } else {
        if (a.kv_bf16) hipLaunchKernelGGL((k_attention<1, true>), grid, dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((k_attention<1, false>), grid, dim3(256), lds, stream, a);
    }
}

// ------------------------------------------------------------------------------------------------
// Mimi front end
// ------------------------------------------------------------------------------------------------
// K13: latent -> mimi projector with emb_std/emb_mean folded in (model.go:226-242,294-303)
// A thread owns one output channel and keeps its weight row in registers; a block walks PROJ_ROWS frames, whose latent rows are
// wave-uniform (scalar loads feeding v_fmac).  The k order of the sum is the reference's.
constexpr int PROJ_ROWS = 16, PROJ_LMAX = 64;
__global__ __launch_bounds__(256) void k_projector(const float* latent, int64_t lat_bstride, const float* wp, const float* bp, int b, int t, int f0, int f1,
                                                   int ldim, int c, float* out) {
    const int oc = blockIdx.x * 256 + threadIdx.x;
    const int nf = f1 - f0;
    const int ocl = min(oc, c - 1);
    float w[PROJ_LMAX];
#pragma unroll
    for (int k = 0; k < PROJ_LMAX; k += 4) {
        if (k < ldim) {   // ldim % 4 == 0 (host)
            const float4 v = *reinterpret_cast<const float4*>(wp + (int64_t)ocl * ldim + k);
            w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
        }
    }
    const float bias = bp[ocl];
    const int row0 = blockIdx.y * PROJ_ROWS, row1 = min(row0 + PROJ_ROWS, b * nf);
    for (int row = row0; row < row1; row++) {
        const int bi = row / nf, f = f0 + row % nf;
        const float* l = latent + (int64_t)bi * lat_bstride + (int64_t)f * ldim;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < PROJ_LMAX; k++)
            if (k < ldim) s += l[k] * w[k];
        if (oc < c) out[((int64_t)bi * (t + 1) + 1 + f) * c + oc] = s + bias;   // row 0 of every utterance is the zero history row
    }
}
__global__ void k_projector_any(const float* latent, int64_t lat_bstride, const float* wp, const float* bp, int b, int t, int f0, int f1,
                                int ldim, int c, float* out) {   // any ldim
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nf = f1 - f0;
    int64_t tot = (int64_t)b * nf * c;
    if (i >= tot) return;
    int oc = (int)(i % c);
    int f = f0 + (int)((i / c) % nf);
    int bi = (int)(i / ((int64_t)c * nf));
    const float* l = latent + (int64_t)bi * lat_bstride + (int64_t)f * ldim;
    const float* w = wp + (int64_t)oc * ldim;
    float s = 0.0f;
    for (int k = 0; k < ldim; k++) s += l[k] * w[k];
    out[((int64_t)bi * (t + 1) + 1 + f) * c + oc] = s + bp[oc];
}
void launch_projector(const float* latent, int64_t lat_bstride, const float* wp, const float* bp, int b, int t, int f0, int f1,
                      int ldim, int c, float* out, hipStream_t stream) {
    int64_t tot = (int64_t)b * (f1 - f0) * c;
    if (tot <= 0) return;
    if (ldim <= PROJ_LMAX && ldim % 4 == 0 && aligned16(wp)) {
        const int rows = b * (f1 - f0);
        hipLaunchKernelGGL(k_projector, dim3((unsigned)((c + 255) / 256), (unsigned)((rows + PROJ_ROWS - 1) / PROJ_ROWS)), dim3(256), 0, stream, latent, lat_bstride,
                           wp, bp, b, t, f0, f1, ldim, c, out);
        return;
    }
    hipLaunchKernelGGL(k_projector_any, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, latent, lat_bstride, wp, bp, b, t, f0, f1, ldim, c, out);
}

// K14: depthwise transposed conv, k = 2*stride, first T*stride outputs kept (convtranspose1d.go:154-202, mimi.go:116-125)
// One block per input frame: a thread owns four channels, reads frame t-1 and frame t once and writes the frame's `stride`
// output rows (no per-element index arithmetic; every access is a coalesced float4).
__global__ __launch_bounds__(128) void k_upsample_dw4(const float* in, const float* w0, const float* w1, const float* bias, int t, int f0, int nf, int c,
                                                     int stride, float* out, int pad) {
    const int bi = blockIdx.x / nf, tt = f0 + blockIdx.x % nf;
    const float* x = in + ((int64_t)bi * (t + 1) + tt) * c;   // x[0..c) = frame t-1 (row 0 is the zero row), x[c..2c) = frame t
    float* o = out + ((int64_t)bi * (pad + (int64_t)t * stride) + pad + (int64_t)tt * stride) * c;
    for (int ch = threadIdx.x * 4; ch < c; ch += 512) {
        const float4 p = *reinterpret_cast<const float4*>(x + ch), q = *reinterpret_cast<const float4*>(x + c + ch);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4*>(bias + ch);
        for (int r = 0; r < stride; r++) {
            const float4 a = *reinterpret_cast<const float4*>(w0 + (int64_t)r * c + ch), d = *reinterpret_cast<const float4*>(w1 + (int64_t)r * c + ch);
            float4 v;
            v.x = p.x * a.x + q.x * d.x; v.y = p.y * a.y + q.y * d.y; v.z = p.z * a.z + q.z * d.z; v.w = p.w * a.w + q.w * d.w;
            if (bias) { v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
            *reinterpret_cast<float4*>(o + (int64_t)r * c + ch) = v;
        }
    }
}
__global__ void k_upsample_dw(const float* in, const float* w0, const float* w1, const float* bias, int b, int t, int f0, int f1, int c,
                              int stride, float* out, int pad) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nf = f1 - f0;
    int64_t tot = (int64_t)b * nf * stride * c;
    if (i >= tot) return;
    int ch = (int)(i % c);
    int r = (int)((i / c) % stride);
    int tt = f0 + (int)((i / ((int64_t)c * stride)) % nf);
    int bi = (int)(i / ((int64_t)c * stride * nf));
    const float* x = in + ((int64_t)bi * (t + 1) + tt) * c + ch;  // x[0] = frame t-1 (row 0 is the zero row), x[c] = frame t
    float prev = x[0] * w0[r * c + ch];
    float cur = x[c] * w1[r * c + ch];
    float v = prev + cur;
    if (bias) v += bias[ch];
    out[((int64_t)bi * (pad + (int64_t)t * stride) + pad + (int64_t)tt * stride + r) * c + ch] = v;
}
void launch_upsample_depthwise(const float* in, const float* w0, const float* w1, const float* bias, int b, int t, int f0, int f1,
                               int c, int stride, float* out, int out_pad_rows, hipStream_t stream) {
    int64_t tot = (int64_t)b * (f1 - f0) * stride * c;
    if (tot <= 0) return;
    if (c % 4 == 0 && aligned16(in) && aligned16(w0) && aligned16(w1) && aligned16(out) && (!bias || aligned16(bias)) && (int64_t)b * (f1 - f0) < (1 << 30)) {
        hipLaunchKernelGGL(k_upsample_dw4, dim3((unsigned)(b * (f1 - f0))), dim3(128), 0, stream, in, w0, w1, bias, t, f0, f1 - f0, c, stride, out, out_pad_rows);
        return;
    }
    hipLaunchKernelGGL(k_upsample_dw, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, in, w0, w1, bias, b, t, f0, f1, c, stride, out, out_pad_rows);
}

template <bool KVBF16>
__global__ void k_voice_scatter(const float* raw, int t, int heads, int hd, int offset, int slot, void* kc, void* vc, int64_t cap) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t tot = (int64_t)offset * heads * hd;   // rows beyond `offset` hold NaN padding and are never read
    if (i >= tot) return;
    int e = (int)(i % hd);
    int h = (int)((i / hd) % heads);
    int step = (int)(i / ((int64_t)hd * heads));
    float kv = raw[((int64_t)step * heads + h) * hd + e];                      // voiceKVIndex(0, ...)
    float vv = raw[(((int64_t)t + step) * heads + h) * hd + e];                // voiceKVIndex(1, ...)
    int64_t dst = (((int64_t)slot * heads + h) * cap + step) * hd + e;
    if (KVBF16) {
        reinterpret_cast<unsigned short*>(kc)[dst] = f32_to_bf16_bits(kv);
        reinterpret_cast<unsigned short*>(vc)[dst] = f32_to_bf16_bits(vv);
    } else {
        reinterpret_cast<float*>(kc)[dst] = kv;
        reinterpret_cast<float*>(vc)[dst] = vv;
    }
}
void launch_voice_scatter(const float* raw, int t, int heads, int hd, int offset, int slot, void* kcache, void* vcache,
                          int kv_bf16, int64_t cap, hipStream_t stream) {
    int64_t tot = (int64_t)offset * heads * hd;
    if (tot <= 0) return;
    dim3 g((unsigned)((tot + 255) / 256));
    if (kv_bf16) hipLaunchKernelGGL(k_voice_scatter<true>, g, dim3(256), 0, stream, raw, t, heads, hd, offset, slot, kcache, vcache, cap);
    else hipLaunchKernelGGL(k_voice_scatter<false>, g, dim3(256), 0, stream, raw, t, heads, hd, offset, slot, kcache, vcache, cap);
}

__global__ void k_voice_apply(const uint4* vk, const uint4* vv, int offset, int heads, int row16, const int32_t* slots, int n_slots,
                              uint4* kc, uint4* vc, int64_t cap) {
    int64_t per_slot = (int64_t)heads * offset * row16;   // 16-byte chunks per slot
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_slot * n_slots) return;
    int si = (int)(i / per_slot);
    int64_t r = i % per_slot;
    int c = (int)(r % row16);
    int row = (int)((r / row16) % offset);
    int h = (int)(r / ((int64_t)row16 * offset));
    int64_t dst = (((int64_t)slots[si] * heads + h) * cap + row) * row16 + c;
    kc[dst] = vk[r];
    vc[dst] = vv[r];
}
void launch_voice_apply(const void* vk, const void* vv, int offset, int heads, int hd, const int32_t* slots, int n_slots,
                        void* kcache, void* vcache, int elem_bytes, int64_t cap, hipStream_t stream) {
    int row16 = hd * elem_bytes / 16;
    int64_t tot = (int64_t)heads * offset * row16 * n_slots;
    if (tot <= 0) return;
    hipLaunchKernelGGL(k_voice_apply, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, (const uint4*)vk, (const uint4*)vv, offset,
                       heads, row16, slots, n_slots, (uint4*)kcache, (uint4*)vcache, cap);
}

// K16 (last layer): ELU then causal conv C -> 1 (mimi.go:781-783).  A block produces 256 consecutive samples of one
// utterance: the (256 + k - 1) x C input window is one contiguous span in the channels-last buffer, staged into LDS with
// coalesced 16-byte loads (ELU applied on the way); each thread then reduces its k*C window from LDS.  Rows are padded
// by one float so that the 64 lanes (64 consecutive rows) read 64 different banks.
__global__ __launch_bounds__(256) void k_conv_final(const float* in, int pad, const float* w, const float* bias, int b, int t, int ta, int tb,
                                                    int c, int k, int elu_in, float* out) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [(256 + k - 1)][c + 1], then w[k * c]
    const int tiles_per_b = (tb - ta + 255) / 256;
    const int bi = blockIdx.x / tiles_per_b, t0 = ta + (blockIdx.x % tiles_per_b) * 256;
    const int rows = min(256, tb - t0) + k - 1;
    const int ld = c + 1;
    float* wl = tile + (256 + k - 1) * ld;
    const float* src = in + ((int64_t)bi * (pad + t) + pad + t0 - (k - 1)) * c;
    const int nv = rows * c / 4;
    for (int i = threadIdx.x; i < nv; i += 256) {
        float4 v = reinterpret_cast<const float4*>(src)[i];
        int r = (i * 4) / c, col = (i * 4) % c;
        float* d = tile + r * ld + col;
        if (elu_in) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int i = threadIdx.x; i < k * c; i += 256) wl[i] = w[i];
    __syncthreads();
    const int tt = t0 + threadIdx.x;
    if (tt >= tb) return;
    float s = 0.0f;
    for (int x = 0; x < k; x++) {
        const float* row = tile + (threadIdx.x + x) * ld;
        const float* wr = wl + x * c;
        for (int j = 0; j < c; j++) s += row[j] * wr[j];
    }
    out[(int64_t)bi * t + tt] = s + (bias ? bias[0] : 0.0f);
}
void launch_conv_final(const float* in, int in_pad_rows, const float* w, const float* bias, int b, int t, int t0, int t1, int c, int k,
                       int elu_in, float* out, hipStream_t stream) {
    const int tiles = (t1 - t0 + 255) / 256;
    if (tiles <= 0) return;
    size_t lds = ((size_t)(256 + k - 1) * (c + 1) + (size_t)k * c) * sizeof(float);
    hipLaunchKernelGGL(k_conv_final, dim3((unsigned)(b * tiles)), dim3(256), lds, stream, in, in_pad_rows, w, bias, b, t, t0, t1, c, k, elu_in, out);
}


// ------------------------------------------------------------------------------------------------
// PCM egress: audio.WritePCM16Samples on the device -- halves the bytes that cross PCIe
// ------------------------------------------------------------------------------------------------
__global__ void k_pcm16(const float* in, int16_t* out, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const float4 a = *reinterpret_cast<const float4*>(in + i), b = *reinterpret_cast<const float4*>(in + i + 4);
        uint4 o;
        o.x = (unsigned)(pcm16_one(a.x) & 0xffff) | ((unsigned)pcm16_one(a.y) << 16);
        o.y = (unsigned)(pcm16_one(a.z) & 0xffff) | ((unsigned)pcm16_one(a.w) << 16);
        o.z = (unsigned)(pcm16_one(b.x) & 0xffff) | ((unsigned)pcm16_one(b.y) << 16);
        o.w = (unsigned)(pcm16_one(b.z) & 0xffff) | ((unsigned)pcm16_one(b.w) << 16);
        *reinterpret_cast<uint4*>(out + i) = o;
    } else {
        for (int64_t j = i; j < n; j++) out[j] = (int16_t)pcm16_one(in[j]);
    }
}
__global__ void k_pcm16_rows(const float* in, int16_t* out, int64_t row_stride, int64_t off, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= n) return;
    const int64_t base = (int64_t)blockIdx.y * row_stride + off + i;
    for (int j = 0; j < 8 && i + j < n; j++) out[base + j] = (int16_t)pcm16_one(in[base + j]);
}
void launch_pcm16_rows(const float* in, int16_t* out, int rows, int64_t row_stride, int64_t off, int64_t n, hipStream_t stream) {
    if (n <= 0 || rows <= 0) return;
    const int64_t threads = (n + 7) / 8;
    hipLaunchKernelGGL(k_pcm16_rows, dim3((unsigned)((threads + 255) / 256), (unsigned)rows), dim3(256), 0, stream, in, out, row_stride, off, n);
}
void launch_pcm16(const float* in, int16_t* out, int64_t n, hipStream_t stream) {
    if (n <= 0) return;
    const int64_t threads = (n + 7) / 8;
    hipLaunchKernelGGL(k_pcm16, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, in, out, n);
}

__global__ void k_zero_rows(float* base, int64_t batch_stride, int b, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)b * n) return;
    base[(i / n) * batch_stride + (i % n)] = 0.0f;
}
void launch_zero_rows(float* base, int64_t batch_stride, int b, int64_t n, hipStream_t stream) {
    int64_t tot = (int64_t)b * n;
    if (tot <= 0) return;
    hipLaunchKernelGGL(k_zero_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, base, batch_stride, b, n);
}

// ------------------------------------------------------------------------------------------------
// AR-step bookkeeping: the per-utterance loop state of runtime_native_safetensors.go:150-201 on device
// ------------------------------------------------------------------------------------------------
// One launch opens the step.  Per utterance: the model input (previous latent, or the BOS marker replaced by bos_emb:
// newBOSSequenceTensor :246-253 + replaceNaNWithVector) and the starting point of the flow (this step's noise or zeros);
// then, when the two 32-wide linears are handed in, what they feed: x = input_linear(input) (flow_lm.go:254) and
// fx = input_proj(start) (flow_net.go:327) -- 32 multiply-adds per output in exact f32, not worth a launch each.
template <bool WBF16>
__global__ __launch_bounds__(256) void k_step_begin(StepState s, const float* latents, int64_t lat_stride, const float* bos, const float* noise,
                                                    int64_t noise_stride, int ldim, float* in32, float* x0, const void* w_in, const float* b_in, int d_in,
                                                    float* x, const void* w_pj, const float* b_pj, int d_pj, float* fx) {
    __shared__ float vin[64], vst[64];
    const int bi = blockIdx.y, tid = threadIdx.x;
    // the thread's weight row is requested first: it does not depend on the step's input, whose own chain (step counter -> latent
    // row -> LDS -> barrier) then runs under that fetch instead of in front of it.  ldim <= 64 (host), ldim % 4 == 0.
    const int n = blockIdx.x * 256 + tid;
    const bool have = w_in && n < d_in + d_pj;
    const bool first = n < d_in;
    const int row = first ? n : n - d_in;
    const char* wrow = (const char*)(first ? w_in : w_pj) + (int64_t)(have ? row : 0) * ldim * (WBF16 ? 2 : 4);
    constexpr int WQ = 16;                              // 4-k groups of a 64-wide row
    uint2 wb[WBF16 ? WQ : 1];
    float4 wf[WBF16 ? 1 : WQ];
    if (w_in) {
#pragma unroll
        for (int q = 0; q < WQ; q++) {
            if (q * 4 < ldim) {
                if constexpr (WBF16) wb[q] = *reinterpret_cast<const uint2*>(wrow + q * 8);
                else wf[q] = *reinterpret_cast<const float4*>(wrow + q * 16);
            }
        }
    }
    if (tid < ldim) {
        const int st = s.step[bi];
        const float v = st == 0 ? NAN : latents[(int64_t)bi * lat_stride + (int64_t)(st - 1) * ldim + tid];
        const float vi = isnan(v) ? bos[tid] : v;
        const float vs = (noise && st < s.max_steps[bi]) ? noise[(int64_t)bi * noise_stride + (int64_t)st * ldim + tid] : 0.0f;   // a finished utterance has no row `st`
        vin[tid] = vi;
        vst[tid] = vs;
        if (blockIdx.x == 0) { in32[bi * ldim + tid] = vi; x0[bi * ldim + tid] = vs; }
    }
    __syncthreads();
    if (!have) return;
    const float* vec = first ? vin : vst;
    const float* bias = first ? b_in : b_pj;
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < WQ; q++) {
        if (q * 4 < ldim) {
            const int k = q * 4;
            float w0, w1, w2, w3;
            if constexpr (WBF16) {
                const uint2 u = wb[q];
                w0 = __uint_as_float(u.x << 16); w1 = __uint_as_float(u.x & 0xffff0000u); w2 = __uint_as_float(u.y << 16); w3 = __uint_as_float(u.y & 0xffff0000u);
            } else {
                const float4 u = wf[q];
                w0 = u.x; w1 = u.y; w2 = u.z; w3 = u.w;
            }
            acc = fmaf(w0, vec[k], acc); acc = fmaf(w1, vec[k + 1], acc); acc = fmaf(w2, vec[k + 2], acc); acc = fmaf(w3, vec[k + 3], acc);
        }
    }
    acc += bias ? bias[row] : 0.0f;
    if (first) x[(int64_t)bi * d_in + row] = acc;
    else fx[(int64_t)bi * d_pj + row] = acc;
}
// The same opening on the matrix core (step_open.h): a block = 16 rows x 64 columns of [x | fx].  Every block builds the bf16 hi / lo planes of its
// rows' input frame and noise row (16 x 32 each), then each of its four waves multiplies one 16-column tile.  Same functions, same operand split as
// the chained form in the step's last launch (skinny.hip): either way a step starts from the same bits.
template <bool WBF16>
__global__ __launch_bounds__(256) void k_step_begin_mm(StepState s, const float* latents, int64_t lat_stride, const float* noise, int64_t noise_stride, int b,
                                                       float* in32, StepChain ch) {
    __shared__ __attribute__((aligned(16))) unsigned char ih[16 * SO_PITCH], il[16 * SO_PITCH], nh[16 * SO_PITCH], nl[16 * SO_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 16;
    // the wave's weight fragment first: it does not depend on the step's input
    int n0 = blockIdx.x * 64 + wave * 16;
    const bool first = n0 < ch.d_in;
    if (!first) n0 -= ch.d_in;
    const bool have = first || n0 < ch.d_pj;
    const SoW<WBF16> w = so_load_w<WBF16>(first ? ch.w_in : ch.w_pj, first ? ch.b_in : ch.b_pj, have ? n0 : 0, first ? ch.d_in : ch.d_pj, lane);
    {
        const int r = tid >> 4, c = (tid & 15) * 2, bi = m0 + r;   // 16 rows x 16 column pairs
        float2 vi = make_float2(0.f, 0.f), vs = make_float2(0.f, 0.f);
        if (bi < b) {
            const int st = s.step[bi];
            if (st == 0) vi = make_float2(NAN, NAN);
            else vi = *reinterpret_cast<const float2*>(latents + (int64_t)bi * lat_stride + (int64_t)(st - 1) * SO_K + c);
            if (isnan(vi.x)) vi.x = ch.bos[c];
            if (isnan(vi.y)) vi.y = ch.bos[c + 1];
            if (noise && st < s.max_steps[bi]) vs = *reinterpret_cast<const float2*>(noise + (int64_t)bi * noise_stride + (int64_t)st * SO_K + c);   // a finished utterance has no row `st`
            if (blockIdx.x == 0) {
                *reinterpret_cast<float2*>(in32 + (int64_t)bi * SO_K + c) = vi;
                *reinterpret_cast<float2*>(ch.x0 + (int64_t)bi * SO_K + c) = vs;
            }
        }
        so_put(ih, il, r, c, vi.x); so_put(ih, il, r, c + 1, vi.y);
        so_put(nh, nl, r, c, vs.x); so_put(nh, nl, r, c + 1, vs.y);
    }
    __syncthreads();
    if (!have) return;
    if (first) so_tile<WBF16>(ih, il, w, n0, ch.x, ch.d_in, m0, b, lane);
    else so_tile<WBF16>(nh, nl, w, n0, ch.fx, ch.d_pj, m0, b, lane);
}

bool step_open_mfma_ok(const StepOpenLinears& lin, int ldim) {
    return ldim == SO_K && lin.d_in > 0 && lin.d_pj > 0 && lin.d_in % 16 == 0 && lin.d_pj % 16 == 0 && aligned16(lin.w_in) && aligned16(lin.w_pj) &&
           (!lin.b_in || aligned16(lin.b_in)) && (!lin.b_pj || aligned16(lin.b_pj)) && aligned16(lin.x) && aligned16(lin.fx);
}

void launch_step_begin(const StepState& s, const float* latents, int64_t lat_stride, const float* bos, const float* noise, int64_t noise_stride,
                       int ldim, int b, float* in32, float* x0, const StepOpenLinears* lin, hipStream_t stream) {
    if (lin && step_open_mfma_ok(*lin, ldim) && lat_stride % 2 == 0 && noise_stride % 2 == 0) {
        StepChain ch{lin->w_in, lin->b_in, lin->x, lin->d_in, lin->w_pj, lin->b_pj, lin->fx, lin->d_pj, bos, x0, lin->w_bf16, 1};
        dim3 grid((unsigned)((lin->d_in + lin->d_pj + 63) / 64), (unsigned)((b + 15) / 16));
        note_launch("k_step_begin_mm");
        if (lin->w_bf16) hipLaunchKernelGGL(k_step_begin_mm<true>, grid, dim3(256), 0, stream, s, latents, lat_stride, noise, noise_stride, b, in32, ch);
        else hipLaunchKernelGGL(k_step_begin_mm<false>, grid, dim3(256), 0, stream, s, latents, lat_stride, noise, noise_stride, b, in32, ch);
        return;
    }
    const int cols = lin ? lin->d_in + lin->d_pj : 0;
    dim3 grid(lin ? (cols + 255) / 256 : 1, b);
    const void* w_in = lin ? lin->w_in : nullptr;
    if (lin && lin->w_bf16)
        hipLaunchKernelGGL(k_step_begin<true>, grid, dim3(256), 0, stream, s, latents, lat_stride, bos, noise, noise_stride, ldim, in32, x0, w_in,
                           lin->b_in, lin->d_in, lin->x, lin->w_pj, lin->b_pj, lin->d_pj, lin->fx);
    else
        hipLaunchKernelGGL(k_step_begin<false>, grid, dim3(256), 0, stream, s, latents, lat_stride, bos, noise, noise_stride, ldim, in32, x0, w_in,
                           lin ? lin->b_in : nullptr, lin ? lin->d_in : 0, lin ? lin->x : nullptr, lin ? lin->w_pj : nullptr,
                           lin ? lin->b_pj : nullptr, lin ? lin->d_pj : 0, lin ? lin->fx : nullptr);
}

// ------------------------------------------------------------------------------------------------
// Sampling noise (FlowLM.makeGaussianNoise, flow_lm.go:386-408): N(0,1) * sqrt(max(temperature, 0)) per (utterance, step, latent
// element).  The reference draws from a math/rand stream seeded with the wall clock when the runtime is created
// (runtime_native_safetensors.go:27-32), i.e. it promises a distribution, not a sequence; here the draw is counter-based --
// Philox-4x32-10 keyed by the request's seed, counter = (step, element quad) -- so that a (seed, step) pair names its noise row
// whatever the batch composition, and Box-Muller turns each 4 x 32 bits into 4 normals.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__global__ void k_noise_fill(const NoiseSpec* spec, float* out, int64_t out_stride, int ldim) {
    const NoiseSpec sp = spec[blockIdx.y];
    const int quads = ldim / 4;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sp.rows * quads) return;
    const int step = i / quads, qd = i % quads;
    uint32_t c[4] = {(uint32_t)step, (uint32_t)qd, 0u, 0u};
    uint32_t k0 = (uint32_t)sp.seed, k1 = (uint32_t)(sp.seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    float v[4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1): 24 bits, never 0
        const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float r = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.28318530717958647692f * u2, &sn, &cs);
        v[2 * h] = r * cs * sp.sigma;
        v[2 * h + 1] = r * sn * sp.sigma;
    }
    *reinterpret_cast<float4*>(out + (int64_t)blockIdx.y * out_stride + (int64_t)step * ldim + qd * 4) = make_float4(v[0], v[1], v[2], v[3]);
}
void launch_noise_fill(const NoiseSpec* spec_dev, int n_slots, int max_rows, float* out, int64_t out_stride, int ldim, hipStream_t stream) {
    if (n_slots <= 0 || max_rows <= 0) return;
    const int per = max_rows * (ldim / 4);
    hipLaunchKernelGGL(k_noise_fill, dim3((per + 255) / 256, n_slots), dim3(256), 0, stream, spec_dev, out, out_stride, ldim);
}

__global__ void k_step_finish(StepState s, const float* frame, const float* eos_logit, int ldim, int b, float* latents,
                              int64_t lat_stride) {
    int bi = blockIdx.x;
    if (bi >= b) return;
    if (!s.active[bi]) return;
    int st = s.step[bi];
    int e = threadIdx.x;
    if (e < ldim) latents[(int64_t)bi * lat_stride + (int64_t)st * ldim + e] = frame[bi * ldim + e];  // latentFrames = append(...)
    if (e == 0) {
        bool is_eos = eos_logit[bi] > s.eos_threshold[bi];   // flow_lm.go:281
        int cd = s.countdown[bi];
        bool done = false;
        if (is_eos && cd < 0) { cd = s.frames_after_eos[bi]; s.eos_step[bi] = st; }   // :178-182
        if (cd >= 0) {                                                                // :184-190
            if (cd == 0) { done = true; s.broke[bi] = 1; }
            else cd--;
        }
        s.countdown[bi] = cd;
        s.n_frames[bi] = st + 1;
        s.step[bi] = st + 1;
        s.kv_len[bi] += 1;
        if (st + 1 >= s.max_steps[bi]) done = true;                                   // for step := range maxSteps
        if (done) { s.active[bi] = 0; atomicSub(s.n_active, 1); }
    }
}
void launch_step_finish(const StepState& s, const float* frame, const float* eos_logit, int ldim, int b, float* latents,
                        int64_t lat_stride, hipStream_t stream) {
    hipLaunchKernelGGL(k_step_finish, dim3(b), dim3(64), 0, stream, s, frame, eos_logit, ldim, b, latents, lat_stride);
}

// ------------------------------------------------------------------------------------------------
// layout helpers for the op-level entry points ([B, C, T] <-> channels-last with history rows)
// ------------------------------------------------------------------------------------------------
__global__ void k_bct_to_btc(const float* in, int b, int c, int t, float* out, int pad) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)b * c * t) return;
    int ch = (int)(i % c);
    int tt = (int)((i / c) % t);
    int bi = (int)(i / ((int64_t)c * t));
    out[((int64_t)bi * (pad + t) + pad + tt) * c + ch] = in[((int64_t)bi * c + ch) * t + tt];
}
void launch_bct_to_btc(const float* in, int b, int c, int t, float* out, int out_pad_rows, hipStream_t stream) {
    int64_t tot = (int64_t)b * c * t;
    hipLaunchKernelGGL(k_bct_to_btc, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, in, b, c, t, out, out_pad_rows);
}
__global__ void k_btc_to_bct(const float* in, int pad, int b, int c, int t, float* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)b * c * t) return;
    int tt = (int)(i % t);
    int ch = (int)((i / t) % c);
    int bi = (int)(i / ((int64_t)c * t));
    out[i] = in[((int64_t)bi * (pad + t) + pad + tt) * c + ch];
}
void launch_btc_to_bct(const float* in, int in_pad_rows, int b, int c, int t, float* out, hipStream_t stream) {
    int64_t tot = (int64_t)b * c * t;
    hipLaunchKernelGGL(k_btc_to_bct, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, in, in_pad_rows, b, c, t, out);
}

}  // namespace ptts
