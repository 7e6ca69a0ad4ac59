// attn_window.hip -- sliding-window self-attention of the Mimi decoder transformer (K15; mimi.go:365-441,
// attention.go:307-484 with context 250: a query at position p sees keys p-249 .. p of its own utterance).
#include <type_traits>

#include "kernels.h"
#include "device_util.h"

namespace ptts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One WAVE owns 32 consecutive query rows of one (utterance, head) and walks the union of their windows (at most
// 32 + context - 1 keys) in tiles of 32 keys with the f32 matrix instruction v_mfma_f32_32x32x2_f32 -- f32 in, f32
// accumulate, so no precision is given up against the reference's float32 arithmetic.  Everything is computed
// TRANSPOSED so that no operand ever has to be re-laid-out through LDS:
//   S^T[key][query] = K * Q^T      A = K tile  (lane&31 = key,  lane>>5 = which half of the 64 head dims),
//                                  B = Q^T     (lane&31 = query, same half); 32 steps, step s pairs dim s with dim 32+s.
//   D layout of the 32x32 result:  lane&31 = query, register r = key (r&3) + 8*(r>>2) + 4*(lane>>5)
//   O^T[dim][query] += V^T * P^T   B at step r is register r of P^T exactly as the first product left it (the two lane
//                                  halves hold the two keys of the step), A = V^T (lane&31 = dim, lane>>5 = key of the pair).
// Each lane therefore owns ONE query column: the running maximum, the running sum and the rescale factor of the
// streaming softmax are per-lane scalars (one cross-half exchange per tile), and the output divides by the sum at
// the end.  Keys are fetched straight from the qkv rows in HBM/L2 (each K row half is 128 contiguous bytes per lane,
// each V fetch is two 128-byte row pieces per instruction); a block is 4 waves = 4 neighbouring query tiles so their
// overlapping windows meet in L1/L2.  Sums are in a fixed order: results are bitwise reproducible.
//
// The same kernel serves the prompt prefill of the FlowLM transformer (flow_transformer.go:749-771; RAGGED): the queries of
// segment s are the packed rows [rag_off[s], rag_off[s+1]) at positions rag_pos0[s] + i, the keys are the segment's cache rows
// 0 .. position (context < 0: no window), stored as f32 or bf16 (KVBF16: widened on load, f32 arithmetic as above).
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

    float q[32];
    {
        const float* qp = a.q + row_off(qm, row) + a.q_col0 + h * 64 + half * 32;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = *reinterpret_cast<const float4*>(qp + i * 4);
            q[4 * i] = t.x * 0.125f; q[4 * i + 1] = t.y * 0.125f; q[4 * i + 2] = t.z * 0.125f; q[4 * i + 3] = t.w * 0.125f;   // 1/sqrt(64), exact
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
        float kr[32];
        {
            const KvT* kp = kb + (int64_t)min(kt + j, p_last) * a.k_row_stride + half * 32;
            if constexpr (KVBF16) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint4 t = *reinterpret_cast<const uint4*>(kp + i * 8);   // 8 bf16: element e is the low half of word e/2 for even e
                    kr[8 * i] = __uint_as_float(t.x << 16); kr[8 * i + 1] = __uint_as_float(t.x & 0xffff0000u);
                    kr[8 * i + 2] = __uint_as_float(t.y << 16); kr[8 * i + 3] = __uint_as_float(t.y & 0xffff0000u);
                    kr[8 * i + 4] = __uint_as_float(t.z << 16); kr[8 * i + 5] = __uint_as_float(t.z & 0xffff0000u);
                    kr[8 * i + 6] = __uint_as_float(t.w << 16); kr[8 * i + 7] = __uint_as_float(t.w & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 t = *reinterpret_cast<const float4*>(kp + i * 4);
                    kr[4 * i] = t.x; kr[4 * i + 1] = t.y; kr[4 * i + 2] = t.z; kr[4 * i + 3] = t.w;
                }
            }
        }
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
        for (int i = 0; i < 32; i++) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[i], q[i], s, 0, 0, 0);
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
        const float alpha = expf(m - m_use);
        m = m_new;
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; r++) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            s[r] = expf(s[r] - m_use);
            l += s[r];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[r], s[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[r], s[r], o1, 0, 0, 0);
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
