// attn_step.hip -- single-token attention of the AR step (K5 + K6 + K7 in one launch).
#include <algorithm>

#include "kernels.h"
#include "device_util.h"

namespace ptts {

// One block = one (utterance, head); 4 waves split that head's keys.  The launch is a chain of dependent memory round
// trips, so the kernel is organised to have as few of them in sequence as possible:
//   1. every kernel argument and the per-utterance scalars (cache length, active flag, shared-prefix pointers) are
//      requested in one batch of scalar loads;
//   2. the burst -- every wave requests ALL the K rows and ALL the V rows it owns (up to 16 + 16 wave-instructions of
//      1 KiB: 8 keys x 128 B in bf16, 4 keys x 256 B in f32) -- is issued BEFORE the step's own q/k/v are touched: the rows
//      already in the cache do not depend on them.  The loads are unconditional (slots past the end re-read the last
//      valid row: finite data, zero weight);
//   3. while the burst is in flight: q and k of this step are rotated by the RoPE table row at the cache offset
//      (rope.go:81-105; position = offset BEFORE the append, flow_transformer.go:340-347), k and v are appended to the
//      cache at that offset and also left in LDS in cache format -- the key slot at the offset takes them from there;
//   4. scores by DPP reductions inside a key's lane group, max and sum through one LDS exchange, P*V partials reduced
//      by DPP inside a row of 16 lanes and then across the 16 rows of the block through LDS, all in a fixed order
//      (bitwise reproducible).
// Keys beyond the offset are never used: they may hold NaN padding of a voice state (attention.go:402-406); an empty key
// set cannot occur (the key at the offset always exists).  Keys j < pre_len come from a shared prefix (a device voice)
// when there is one: identical for every utterance of that voice, so the batch reads one L2-resident copy.
// Handles up to ATT_NI * 4 * (keys per instruction) keys = 512 (bf16) / 256 (f32); longer caches use k_attention.
// NI (template): wave-instructions per operand actually issued -- the host knows an upper bound on the cache length of the
// launch (AttnArgs::keys_now: lengths after the prefill + steps taken), so a step at 200 keys issues 8 + 8 loads per wave
// and the arithmetic for them instead of the 16 + 16 a 512-key cache would need (slots past the end cost a load each).
constexpr int ATT_NI = 16;

// address-space-qualified views: a pointer that arrives through memory is 'generic' to the compiler, which then emits
// flat_load (LDS-aperture check, both wait counters); these say where the data is.  The constant view of a wave-uniform
// index turns into a scalar load.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr_u4;
__device__ __forceinline__ uint4 gload16(const char* p) {
    const u32x4 t = *(gptr_u4)p;
    return make_uint4(t[0], t[1], t[2], t[3]);
}
template <class T> using cptr = const __attribute__((address_space(4))) T*;

// The five leading pointers are copies of a.seg_len / a.active / a.pre_len / a.pre_k / a.pre_v: built with
// -amdgpu-kernarg-preload-count they arrive in SGPRs with the dispatch, so the per-utterance scalar loads below leave at once
// instead of behind the first round trip for the argument block.
template <bool KVBF16, int NI>
__global__ __launch_bounds__(256) void k_attn_step(const int32_t* p_seg_len, const int32_t* p_active, const int32_t* p_pre_len, const void* const* p_pre_k,
                                                   const void* const* p_pre_v, AttnArgs a) {
    constexpr int LPK = KVBF16 ? 8 : 16;       // lanes per key (16 B each)
    constexpr int KPI = 64 / LPK;              // keys per wave-instruction
    constexpr int DPL = 64 / LPK;              // head dims per lane (8 or 4)
    constexpr int ES = KVBF16 ? 2 : 4;
    asm volatile("" ::"s"(a.k), "s"(a.v), "s"(a.k_seg_stride), "s"(a.k_head_stride), "s"(a.out), "s"(a.out_ld), "s"(a.qkv), "s"(a.qkv_ld), "s"(a.d_model),
                 "s"(a.cos_t), "s"(a.sin_t), "s"(a.layer), "s"(a.heads));
    __shared__ float qs[64];
    __shared__ __attribute__((aligned(16))) unsigned char own_k[64 * ES], own_v[64 * ES];   // this step's k, v in cache format
    __shared__ __attribute__((aligned(16))) float red_m[16];
    __shared__ float red_l[16];
    __shared__ __attribute__((aligned(16))) float red_o[16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x, seg = blockIdx.y;
    // per-utterance scalars: one batch of scalar loads
    const int pos = ((cptr<int32_t>)p_seg_len)[seg];
    const int live = ((cptr<int32_t>)p_active)[seg];
    const int pre = ((cptr<int32_t>)p_pre_len)[seg];
    const char* pk0 = (const char*)((cptr<const void*>)p_pre_k)[seg];
    const char* pv0 = (const char*)((cptr<const void*>)p_pre_v)[seg];
    asm volatile("" ::"s"(pos), "s"(live), "s"(pre), "s"(pk0), "s"(pv0));
    char* kbase = (char*)a.k + ((int64_t)seg * a.k_seg_stride + (int64_t)h * a.k_head_stride) * ES;
    char* vbase = (char*)a.v + ((int64_t)seg * a.k_seg_stride + (int64_t)h * a.k_head_stride) * ES;
    float* outp = a.out + (int64_t)seg * a.out_ld + h * 64;
    if (!live) {   // uniform per block
        if (tid < 64) outp[tid] = 0.0f;
        return;
    }
    const int64_t pre_off = ((int64_t)a.layer * a.heads + h) * pre * 64 * ES;
    const char* pk = pre ? pk0 + pre_off : kbase;
    const char* pv = pre ? pv0 + pre_off : vbase;
    const int sub = lane % LPK, kq = lane / LPK;
    // ---- this step's q, k, v and the RoPE row: requested first (every lane, so that no load sits under a branch), consumed
    // while the burst is in flight -- s_waitcnt counts loads in issue order, so what is needed first is asked for first ----
    const float* qr = a.qkv + (int64_t)seg * a.qkv_ld + h * 64;
    const int t32 = tid & 31, e64 = (tid + 32) & 63;
    const float rc = a.cos_t[(int64_t)pos * 32 + t32], rs = a.sin_t[(int64_t)pos * 32 + t32];
    const float2 q2 = *reinterpret_cast<const float2*>(qr + 2 * t32);
    const float2 k2 = *reinterpret_cast<const float2*>(qr + a.d_model + 2 * t32);
    const float vv = qr[2 * a.d_model + e64];
    __builtin_amdgcn_sched_barrier(0);
    // ---- burst: all K and V rows of this wave that are already in memory (keys < pos) ----
    uint4 kr[NI], vr[NI];
    const int last = max(pos - 1, 0);
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int jj = min((i * 4 + wave) * KPI + kq, last);
        kr[i] = gload16((jj < pre ? pk : kbase) + ((int64_t)jj * 64 + sub * DPL) * ES);
        vr[i] = gload16((jj < pre ? pv : vbase) + ((int64_t)jj * 64 + sub * DPL) * ES);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- RoPE + append ----
    if (tid < 32) {
        qs[2 * tid] = q2.x * rc - q2.y * rs;
        qs[2 * tid + 1] = q2.x * rs + q2.y * rc;
        const float r0 = k2.x * rc - k2.y * rs, r1 = k2.x * rs + k2.y * rc;
        const int64_t dst = (int64_t)pos * 64 + 2 * tid;
        if (KVBF16) {
            const unsigned pkd = (unsigned)f32_to_bf16_bits(r0) | ((unsigned)f32_to_bf16_bits(r1) << 16);
            *reinterpret_cast<unsigned*>(kbase + dst * 2) = pkd;
            *reinterpret_cast<unsigned*>(own_k + 4 * tid) = pkd;
        } else {
            *reinterpret_cast<float2*>(kbase + dst * 4) = make_float2(r0, r1);
            *reinterpret_cast<float2*>(own_k + 8 * tid) = make_float2(r0, r1);
        }
    } else if (tid < 96) {
        const int e = tid - 32;
        const int64_t dst = (int64_t)pos * 64 + e;
        if (KVBF16) {
            const unsigned short b = f32_to_bf16_bits(vv);
            *reinterpret_cast<unsigned short*>(vbase + dst * 2) = b;
            *reinterpret_cast<unsigned short*>(own_v + 2 * e) = b;
        } else {
            *reinterpret_cast<float*>(vbase + dst * 4) = vv;
            *reinterpret_cast<float*>(own_v + 4 * e) = vv;
        }
    }
    __syncthreads();
    const int nk = pos + 1;
    {   // the key slot at the offset takes this step's k, v from LDS (one lane group of one wave; wave-uniform branches)
        const int g_p = pos / KPI, i_p = g_p >> 2, w_p = g_p & 3, kq_p = pos % KPI;
        if (wave == w_p) {
            const uint4 ok = *reinterpret_cast<const uint4*>(own_k + sub * 16);
            const uint4 ov4 = *reinterpret_cast<const uint4*>(own_v + sub * 16);
#pragma unroll
            for (int i = 0; i < NI; i++) {
                if (i == i_p && kq == kq_p) { kr[i] = ok; vr[i] = ov4; }
            }
        }
    }
    float qv[DPL];
#pragma unroll
    for (int e = 0; e < DPL; e++) qv[e] = qs[sub * DPL + e];
    // ---- scores ----
    float sc[NI];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; i++) {
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
        p += dpp_f32<0xB1>(p);                    // lanes ^1
        p += dpp_f32<0x4E>(p);                    // lanes ^2
        p += dpp_f32<0x141>(p);                   // the other quad of the 8-lane half
        if (LPK == 16) p += dpp_f32<0x140>(p);    // the other half of the row
        const int jj = (i * 4 + wave) * KPI + kq;
        sc[i] = jj < nk ? p * 0.125f : -INFINITY;   // 1/sqrt(64)
        mx = fmaxf(mx, sc[i]);
    }
    if (LPK == 8) mx = fmaxf(mx, dpp_f32<0x128>(mx));   // the other key group of the row (row_ror:8)
    if ((lane & 15) == 0) red_m[wave * 4 + (lane >> 4)] = mx;
    __syncthreads();
    {
        const float4 m0 = *reinterpret_cast<const float4*>(red_m), m1 = *reinterpret_cast<const float4*>(red_m + 4);
        const float4 m2 = *reinterpret_cast<const float4*>(red_m + 8), m3 = *reinterpret_cast<const float4*>(red_m + 12);
        mx = fmaxf(fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w))),
                   fmaxf(fmaxf(fmaxf(m2.x, m2.y), fmaxf(m2.z, m2.w)), fmaxf(fmaxf(m3.x, m3.y), fmaxf(m3.z, m3.w))));   // finite: the key at `pos` exists
    }
    // ---- P * V ----
    float ov[DPL];
#pragma unroll
    for (int e = 0; e < DPL; e++) ov[e] = 0.0f;
    float l = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const float p = __expf(sc[i] - mx);   // exp(-inf) = 0 for slots past the end (their V rows are valid, finite data)
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
    // every lane of a key group holds the same p: l is per key group.  Reduce the key groups of a row (bf16: two per row),
    // then the 16 rows of the block through LDS.
    if (LPK == 8) {
        l += dpp_f32<0x128>(l);
#pragma unroll
        for (int e = 0; e < DPL; e++) ov[e] += dpp_f32<0x128>(ov[e]);
    }
    {
        const int row = wave * 4 + (lane >> 4);
        if ((lane & 15) < LPK) {
#pragma unroll
            for (int e = 0; e < DPL; e++) red_o[row][sub * DPL + e] = ov[e];
        }
        if ((lane & 15) == 0) red_l[row] = l;
    }
    __syncthreads();
    if (tid < 64) {
        float den = 0.0f, num = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) { den += red_l[r]; num += red_o[r][tid]; }
        const float o = num / den;
        outp[tid] = o;
    }
}

bool attn_step_supported(const AttnArgs& a) {
    const int kpi = a.kv_bf16 ? 8 : 4;
    return a.fused_step && a.hd == 64 && a.context < 0 && a.max_keys <= ATT_NI * 4 * kpi && a.k_row_stride == 64 && a.active && a.seg_len &&
           a.pre_len && a.pre_k && a.pre_v && a.qkv_ld % 2 == 0 && a.d_model % 2 == 0;
}

template <bool KVBF16, int NI>
static void launch_ni(const AttnArgs& a, int ni, dim3 grid, hipStream_t stream) {   // the smallest instantiation that covers ni rounds
    if constexpr (NI < ATT_NI) {
        if (ni > NI) { launch_ni<KVBF16, NI + 1>(a, ni, grid, stream); return; }
    }
    hipLaunchKernelGGL((k_attn_step<KVBF16, NI>), grid, dim3(256), 0, stream, a.seg_len, a.active, a.pre_len, a.pre_k, a.pre_v, a);
}

int attn_step_keys_per_round(bool kv_bf16) { return 4 * (kv_bf16 ? 8 : 4); }
int attn_step_rounds(int keys, bool kv_bf16) {
    const int kpw = attn_step_keys_per_round(kv_bf16);
    return std::max(1, std::min(ATT_NI, (keys + kpw - 1) / kpw));
}

void launch_attn_step(const AttnArgs& a, hipStream_t stream) {
    note_launch("k_attn_step");
    dim3 grid(a.heads, a.rows);
    const int keys = a.keys_now > 0 ? std::min(a.keys_now, a.max_keys) : a.max_keys;   // unknown: the whole cache
    const int ni = attn_step_rounds(keys, a.kv_bf16 != 0);
    if (a.kv_bf16) launch_ni<true, 1>(a, ni, grid, stream);
    else launch_ni<false, 1>(a, ni, grid, stream);
}

}  // namespace ptts
