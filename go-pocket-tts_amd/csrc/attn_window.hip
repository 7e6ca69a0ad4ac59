// attn_window.hip -- sliding-window self-attention of the Mimi decoder transformer (K15; mimi.go:365-441,
// attention.go:307-484 with context 250: a query at position p sees keys p-249 .. p of its own utterance).
#include <type_traits>

#include "kernels.h"
#include "device_util.h"

namespace ptts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

union FragW {
    bf16x8 v;
    uint4 q;
    unsigned u[4];
};

// (a, b) -> packed bf16 hi pair and bf16 lo pair with a = hi + lo to ~2^-17 (round-to-nearest-even)
__device__ __forceinline__ void split2w(float a, float b, unsigned& hi, unsigned& lo) {
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);
    f32x2 r = f - __builtin_convertvector(h, f32x2);
    bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

// One WAVE owns 32 consecutive query rows of one (utterance, head) and walks the union of their windows (at most
// 32 + context - 1 keys) in tiles of 32 keys on the matrix cores.  Everything is computed TRANSPOSED so that no operand
// ever has to be re-laid-out through LDS:
//   S^T[key][query] = K * Q^T      A = K tile  (lane&31 = key,   lane>>5 = k group),
//                                  B = Q^T     (lane&31 = query, lane>>5 = k group); v_mfma_f32_32x32x16_bf16, step i covers head
//                                  dims 16 i .. 16 i + 15, k group g of a step = dims 16 i + 8 g .. + 7 (32 contiguous bytes per lane).
//   D layout of the 32x32 result:  lane&31 = query, register r = key (r&3) + 8*(r>>2) + 4*(lane>>5)
//   O^T[dim][query] += V^T * P^T   B of step i = registers 8 i .. 8 i + 7 of P^T exactly as the first product left them (the k
//                                  index of the second product is DEFINED as that order: k group g, element e of step i = key
//                                  (r&3) + 8*(r>>2) + 4 g with r = 8 i + e), A = V^T (lane&31 = dim, same keys).
// Operands are f32 split into bf16 hi + lo halves in registers (x = hi + lo to ~2^-17) and every product is three MFMAs
// (hi*hi + lo*hi + hi*lo, f32 accumulation): products exact to ~2^-16 relative -- an eighth of the matrix-core time of the f32
// instruction (v_mfma_f32_32x32x2_f32), which this kernel used first (1090 -> ~600 us per Mimi layer at batch 64).  A bf16 KV cache needs no split
// (its lo half is zero): two MFMAs per product.
// Each lane owns ONE query column: the running maximum, the running sum and the rescale factor of the streaming softmax are
// per-lane scalars (one cross-half exchange per tile), and the output divides by the sum at the end.  Keys are fetched
// straight from the qkv rows in HBM/L2; a block is 4 waves = 4 neighbouring query tiles so their overlapping windows meet in
// L1/L2.  Sums are in a fixed order: results are bitwise reproducible.
//
// The same kernel serves the prompt prefill of the FlowLM transformer (flow_transformer.go:749-771; RAGGED): the queries of
// segment s are the packed rows [rag_off[s], rag_off[s+1]) at positions rag_pos0[s] + i, the keys are the segment's cache rows
// 0 .. position (context < 0: no window), stored as f32 or bf16 (KVBF16).
template <bool KVBF16, bool RAGGED>
__global__ __launch_bounds__(256) void k_attn_window(AttnArgs a, int qtiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.x;
    const int groups = (qtiles + 3) >> 2;
    const int seg = blockIdx.y / groups, qt = (blockIdx.y % groups) * 4 + wave;
    if (qt >= qtiles) return;   // whole wave; the kernel has no block-level synchronisation
    const int r0 = qt * 32;
    const int seg_row0 = RAGGED ? a.rag_off[seg] : seg * a.rows_per_seg;
    const int seg_rows = RAGGED ? a.rag_off[seg + 1] - seg_row0 : a.rows_per_seg;
    if (r0 >= seg_rows) return;
    const int nq = min(32, seg_rows - r0);
    const int j = lane & 31, half = lane >> 5;
    const int my_q = min(j, nq - 1);                 // lanes past a ragged end replay the last query and store nothing
    const int p_first = (RAGGED ? a.rag_pos0[seg] : a.pos_base) + r0, p_last = p_first + nq - 1;
    const int my_pos = p_first + my_q;
    const int ctx = a.context > 0 ? a.context : (1 << 30);
    const int j_lo = max(0, p_first - ctx + 1);
    const RowMap qm{a.q_ld, a.q_rows_per_batch, a.q_batch_stride}, om{a.out_ld, a.o_rows_per_batch, a.o_batch_stride};
    const int row = seg_row0 + r0 + my_q;

    FragW qh[4], ql[4];   // step i: dims 16 i + 8 half .. + 7 of the lane's query, scaled by 1/sqrt(64) (exact)
    {
        const float* qp = a.q + row_off(qm, row) + a.q_col0 + h * 64 + half * 8;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float4 t0 = *reinterpret_cast<const float4*>(qp + 16 * i), t1 = *reinterpret_cast<const float4*>(qp + 16 * i + 4);
            split2w(t0.x * 0.125f, t0.y * 0.125f, qh[i].u[0], ql[i].u[0]);
            split2w(t0.z * 0.125f, t0.w * 0.125f, qh[i].u[1], ql[i].u[1]);
            split2w(t1.x * 0.125f, t1.y * 0.125f, qh[i].u[2], ql[i].u[2]);
            split2w(t1.z * 0.125f, t1.w * 0.125f, qh[i].u[3], ql[i].u[3]);
        }
    }
    typedef typename std::conditional<KVBF16, unsigned short, float>::type KvT;
    const KvT* kb = (const KvT*)a.k + (int64_t)seg * a.k_seg_stride + (int64_t)h * a.k_head_stride;
    const KvT* vb = (const KvT*)a.v + (int64_t)seg * a.k_seg_stride + (int64_t)h * a.k_head_stride;
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; r++) { o0[r] = 0.0f; o1[r] = 0.0f; }
    float m = -INFINITY, l = 0.0f;

    for (int kt = j_lo; kt <= p_last; kt += 32) {
        FragW kh[4], kl[4];
        {
            const KvT* kp = kb + (int64_t)min(kt + j, p_last) * a.k_row_stride + half * 8;
            if constexpr (KVBF16) {
#pragma unroll
                for (int i = 0; i < 4; i++) kh[i].q = *reinterpret_cast<const uint4*>(kp + 16 * i);   // 8 bf16 = one operand, as stored
            } else {
                float4 t[4][2];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    t[i][0] = *reinterpret_cast<const float4*>(kp + 16 * i);
                    t[i][1] = *reinterpret_cast<const float4*>(kp + 16 * i + 4);
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    split2w(t[i][0].x, t[i][0].y, kh[i].u[0], kl[i].u[0]);
                    split2w(t[i][0].z, t[i][0].w, kh[i].u[1], kl[i].u[1]);
                    split2w(t[i][1].x, t[i][1].y, kh[i].u[2], kl[i].u[2]);
                    split2w(t[i][1].z, t[i][1].w, kh[i].u[3], kl[i].u[3]);
                }
            }
        }
        // values of the tile's 32 keys for dims j and 32 + j, in the key order of the score registers
        float v0[16], v1[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int key = min(kt + (r & 3) + 8 * (r >> 2) + 4 * half, p_last);
            const KvT* vp = vb + (int64_t)key * a.k_row_stride + j;
            if constexpr (KVBF16) {
                v0[r] = __uint_as_float((unsigned)vp[0] << 16);
                v1[r] = __uint_as_float((unsigned)vp[32] << 16);
            } else {
                v0[r] = vp[0];
                v1[r] = vp[32];
            }
        }
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[i].v, qh[i].v, s, 0, 0, 0);
            if constexpr (!KVBF16) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[i].v, qh[i].v, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[i].v, ql[i].v, s, 0, 0, 0);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int key = kt + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool ok = key <= my_pos && key > my_pos - ctx;
            s[r] = ok ? s[r] : -INFINITY;
            tmax = fmaxf(tmax, s[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, WAVE));
        const float m_new = fmaxf(m, tmax);
        const float m_use = m_new == -INFINITY ? 0.0f : m_new;   // a tile may hold no visible key for this query yet
        const float alpha = __expf(m - m_use);   // v_exp_f32 (|rel. error| ~2e-7, as in the step attention): libm's expf costs ~12 instructions here
        m = m_new;
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; r++) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            s[r] = __expf(s[r] - m_use);
            l += s[r];
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            FragW ph, pl, vh0, vl0, vh1, vl1;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                split2w(s[8 * i + 2 * e], s[8 * i + 2 * e + 1], ph.u[e], pl.u[e]);
                split2w(v0[8 * i + 2 * e], v0[8 * i + 2 * e + 1], vh0.u[e], vl0.u[e]);
                split2w(v1[8 * i + 2 * e], v1[8 * i + 2 * e + 1], vh1.u[e], vl1.u[e]);
            }
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh0.v, ph.v, o0, 0, 0, 0);
            if constexpr (!KVBF16) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl0.v, ph.v, o0, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh0.v, pl.v, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh1.v, ph.v, o1, 0, 0, 0);
            if constexpr (!KVBF16) o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl1.v, ph.v, o1, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh1.v, pl.v, o1, 0, 0, 0);
        }
    }
    l += __shfl_xor(l, 32, WAVE);
    if (j < nq) {
        const float inv = 1.0f / l;   // the key at the query's own position is always visible
        float* op = a.out + row_off(om, row) + h * 64 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            *reinterpret_cast<float4*>(op + 8 * g) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *reinterpret_cast<float4*>(op + 32 + 8 * g) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
}

bool attn_window_supported(const AttnArgs& a) {
    const bool ragged = a.rag_off != nullptr;
    const int kalign = a.kv_bf16 ? 8 : 4;   // 16-byte key loads
    if (a.fused_step || a.hd != 64 || a.rows_per_seg <= 0 || a.seg_len || a.active || a.k_row_stride % kalign || a.k_head_stride % kalign ||
        a.k_seg_stride % kalign || a.q_ld % 4 || a.out_ld % 4 || !aligned16(a.k) || !aligned16(a.q) || !aligned16(a.out))
        return false;
    if (ragged) return a.rag_pos0 && a.rag_segs > 0 && a.q_rows_per_batch == 0 && a.o_rows_per_batch == 0;   // packed rows
    return !a.kv_bf16 && a.context > 0 && !a.row_seg && !a.row_pos && a.rows % a.rows_per_seg == 0;
}

void launch_attn_window(const AttnArgs& a, hipStream_t stream) {
    note_launch(a.rag_off ? "k_attn_window<ragged>" : "k_attn_window");
    const int qtiles = (a.rows_per_seg + 31) / 32;   // ragged: rows_per_seg is the longest segment
    const int segs = a.rag_off ? a.rag_segs : a.rows / a.rows_per_seg;
    dim3 grid(a.heads, (unsigned)(segs * ((qtiles + 3) / 4)));
    if (a.rag_off) {
        if (a.kv_bf16) hipLaunchKernelGGL((k_attn_window<true, true>), grid, dim3(256), 0, stream, a, qtiles);
        else hipLaunchKernelGGL((k_attn_window<false, true>), grid, dim3(256), 0, stream, a, qtiles);
    } else {
        hipLaunchKernelGGL((k_attn_window<false, false>), grid, dim3(256), 0, stream, a, qtiles);
    }
}

}  // namespace ptts
