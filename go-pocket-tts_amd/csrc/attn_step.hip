// attn_step.hip -- single-token attention of the AR step (K5 + K6 + K7 in one launch).
#include "kernels.h"
#include "device_util.h"

namespace ptts {

// One block = one (utterance, head); 4 waves split that head's keys.
//   prologue : q and k of this step are rotated by the RoPE table row at the cache offset (rope.go:81-105; position =
//              offset BEFORE the append, flow_transformer.go:340-347) and k, v are appended to the cache at that offset;
//   burst    : every wave requests ALL the K rows and ALL the V rows it owns before consuming any (up to 16 + 16
//              wave-instructions of 1 KiB: 8 keys x 128 B in bf16, 4 keys x 256 B in f32) -- the cache is read once,
//              sequentially, at full line width, and the HBM latency is paid once per launch;
//   softmax  : scores by 8-/16-lane shuffles, max and sum through one LDS exchange, P*V partials combined in a fixed
//              order (bitwise reproducible).  Keys beyond the offset are never touched: they may hold NaN padding of a
//              voice state (attention.go:402-406); an empty key set gives zeros (attention.go:423-425).
// Handles up to ATT_NI * 4 * (keys per instruction) keys = 512 (bf16) / 256 (f32); longer caches use k_attention.
constexpr int ATT_NI = 16;

template <bool KVBF16>
__global__ __launch_bounds__(256) void k_attn_step(AttnArgs a) {
    constexpr int LPK = KVBF16 ? 8 : 16;       // lanes per key (16 B each)
    constexpr int KPI = 64 / LPK;              // keys per wave-instruction
    constexpr int DPL = 64 / LPK;              // head dims per lane (8 or 4)
    __shared__ float qs[64];
    __shared__ float red_m[4], red_l[4];
    __shared__ __attribute__((aligned(16))) float red_o[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, seg = blockIdx.y;
    const int pos = a.seg_len[seg];
    const bool live = !a.active || a.active[seg];
    char* kbase = (char*)a.k + ((int64_t)seg * a.k_seg_stride + (int64_t)h * a.k_head_stride) * (KVBF16 ? 2 : 4);
    char* vbase = (char*)a.v + ((int64_t)seg * a.k_seg_stride + (int64_t)h * a.k_head_stride) * (KVBF16 ? 2 : 4);
    float* outp = a.out + (int64_t)seg * a.out_ld + h * 64;
    // shared prefix (speed only: the rows are also in the cache): identical for every utterance of a voice, so the batch
    // reads one L2-resident copy instead of B private ones from HBM
    const int pre = a.pre_len ? a.pre_len[seg] : 0;
    const int64_t pre_off = ((int64_t)a.layer * a.heads + h) * pre * 64 * (KVBF16 ? 2 : 4);
    const char* pk = pre ? (const char*)a.pre_k[seg] + pre_off : kbase;
    const char* pv = pre ? (const char*)a.pre_v[seg] + pre_off : vbase;
    if (!live) {   // uniform per block
        if (tid < 64) outp[tid] = 0.0f;
        return;
    }
    // ---- RoPE + append ----
    const float* qr = a.qkv + (int64_t)seg * a.qkv_ld + h * 64;
    if (tid < 32) {
        const float c = a.cos_t[(int64_t)pos * 32 + tid], s = a.sin_t[(int64_t)pos * 32 + tid];
        const float q0 = qr[2 * tid], q1 = qr[2 * tid + 1];
        qs[2 * tid] = q0 * c - q1 * s;
        qs[2 * tid + 1] = q0 * s + q1 * c;
        const float k0 = qr[a.d_model + 2 * tid], k1 = qr[a.d_model + 2 * tid + 1];
        const float r0 = k0 * c - k1 * s, r1 = k0 * s + k1 * c;
        const int64_t dst = (int64_t)pos * 64 + 2 * tid;
        if (KVBF16) *reinterpret_cast<unsigned*>(kbase + dst * 2) = (unsigned)f32_to_bf16_bits(r0) | ((unsigned)f32_to_bf16_bits(r1) << 16);
        else *reinterpret_cast<float2*>(kbase + dst * 4) = make_float2(r0, r1);
    } else if (tid < 96) {
        const int e = tid - 32;
        const float vv = qr[2 * a.d_model + e];
        const int64_t dst = (int64_t)pos * 64 + e;
        if (KVBF16) *reinterpret_cast<unsigned short*>(vbase + dst * 2) = f32_to_bf16_bits(vv);
        else *reinterpret_cast<float*>(vbase + dst * 4) = vv;
    }
    __syncthreads();
    const int nk = pos + 1;
    const int sub = lane % LPK, kq = lane / LPK;
    // ---- burst: all K and V rows of this wave ----
    uint4 kr[ATT_NI], vr[ATT_NI];
#pragma unroll
    for (int i = 0; i < ATT_NI; i++) {
        const int jj = (i * 4 + wave) * KPI + kq;
        if (jj < nk) {
            kr[i] = *reinterpret_cast<const uint4*>((jj < pre ? pk : kbase) + ((int64_t)jj * 64 + sub * DPL) * (KVBF16 ? 2 : 4));
            vr[i] = *reinterpret_cast<const uint4*>((jj < pre ? pv : vbase) + ((int64_t)jj * 64 + sub * DPL) * (KVBF16 ? 2 : 4));
        } else {
            kr[i] = make_uint4(0, 0, 0, 0);
            vr[i] = make_uint4(0, 0, 0, 0);
        }
    }
    float qv[DPL];
#pragma unroll
    for (int e = 0; e < DPL; e++) qv[e] = qs[sub * DPL + e];
    // ---- scores ----
    float sc[ATT_NI];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < ATT_NI; i++) {
        float p;
        if (KVBF16) {
            p = qv[0] * __uint_as_float(kr[i].x << 16) + qv[1] * __uint_as_float(kr[i].x & 0xffff0000u) +
                qv[2] * __uint_as_float(kr[i].y << 16) + qv[3] * __uint_as_float(kr[i].y & 0xffff0000u) +
                qv[4] * __uint_as_float(kr[i].z << 16) + qv[5] * __uint_as_float(kr[i].z & 0xffff0000u) +
                qv[6] * __uint_as_float(kr[i].w << 16) + qv[7] * __uint_as_float(kr[i].w & 0xffff0000u);
        } else {
            p = qv[0] * __uint_as_float(kr[i].x) + qv[1] * __uint_as_float(kr[i].y) + qv[2] * __uint_as_float(kr[i].z) +
                qv[3] * __uint_as_float(kr[i].w);
        }
#pragma unroll
        for (int o = LPK / 2; o > 0; o >>= 1) p += __shfl_xor(p, o, WAVE);
        const int jj = (i * 4 + wave) * KPI + kq;
        sc[i] = jj < nk ? p * 0.125f : -INFINITY;   // 1/sqrt(64)
        mx = fmaxf(mx, sc[i]);
    }
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, WAVE));
    if (lane == 0) red_m[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));   // finite: the key at `pos` always exists
    // ---- P * V ----
    float ov[DPL];
#pragma unroll
    for (int e = 0; e < DPL; e++) ov[e] = 0.0f;
    float l = 0.0f;
#pragma unroll
    for (int i = 0; i < ATT_NI; i++) {
        const float p = expf(sc[i] - mx);   // exp(-inf) = 0 for slots past the end
        l += p;
        if (KVBF16) {
            ov[0] += p * __uint_as_float(vr[i].x << 16); ov[1] += p * __uint_as_float(vr[i].x & 0xffff0000u);
            ov[2] += p * __uint_as_float(vr[i].y << 16); ov[3] += p * __uint_as_float(vr[i].y & 0xffff0000u);
            ov[4] += p * __uint_as_float(vr[i].z << 16); ov[5] += p * __uint_as_float(vr[i].z & 0xffff0000u);
            ov[6] += p * __uint_as_float(vr[i].w << 16); ov[7] += p * __uint_as_float(vr[i].w & 0xffff0000u);
        } else {
            ov[0] += p * __uint_as_float(vr[i].x); ov[1] += p * __uint_as_float(vr[i].y);
            ov[2] += p * __uint_as_float(vr[i].z); ov[3] += p * __uint_as_float(vr[i].w);
        }
    }
    // every lane of a key group holds the same p, so l is replicated LPK times inside a group: reduce over groups only
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) {
        l += __shfl_xor(l, o, WAVE);
#pragma unroll
        for (int e = 0; e < DPL; e++) ov[e] += __shfl_xor(ov[e], o, WAVE);
    }
    if (kq == 0) {
#pragma unroll
        for (int e = 0; e < DPL; e++) red_o[wave][sub * DPL + e] = ov[e];
        if (sub == 0) red_l[wave] = l;
    }
    __syncthreads();
    if (tid < 64) {
        const float den = (red_l[0] + red_l[1]) + (red_l[2] + red_l[3]);
        const float num = (red_o[0][tid] + red_o[1][tid]) + (red_o[2][tid] + red_o[3][tid]);
        outp[tid] = num / den;
    }
}

bool attn_step_supported(const AttnArgs& a) {
    const int kpi = a.kv_bf16 ? 8 : 4;
    return a.fused_step && a.hd == 64 && a.context < 0 && a.max_keys <= ATT_NI * 4 * kpi && a.k_row_stride == 64;
}

void launch_attn_step(const AttnArgs& a, hipStream_t stream) {
    dim3 grid(a.heads, a.rows);
    if (a.kv_bf16) hipLaunchKernelGGL(k_attn_step<true>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(k_attn_step<false>, grid, dim3(256), 0, stream, a);
}

}  // namespace ptts
