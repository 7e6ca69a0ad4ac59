// attn_window.hip -- sliding-window self-attention of the Mimi decoder transformer (K15; mimi.go:365-441,
// attention.go:307-484 with context 250: a query at position p sees keys p-249 .. p of its own utterance).
#include <cstdlib>
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
            if constexpr (!KVBF16) PTTS_LO_MFMA(s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[i].v, qh[i].v, s, 0, 0, 0));
            PTTS_LO_MFMA(s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[i].v, ql[i].v, s, 0, 0, 0));
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
            if constexpr (!KVBF16) PTTS_LO_MFMA(o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl0.v, ph.v, o0, 0, 0, 0));
            PTTS_LO_MFMA(o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh0.v, pl.v, o0, 0, 0, 0));
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh1.v, ph.v, o1, 0, 0, 0);
            if constexpr (!KVBF16) PTTS_LO_MFMA(o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl1.v, ph.v, o1, 0, 0, 0));
            PTTS_LO_MFMA(o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh1.v, pl.v, o1, 0, 0, 0));
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

// ------------------------------------------------------------------------------------------------
// The same attention for the decoder's dense case (f32 K / V, a window, every segment the same length) with the key tiles SHARED: a block is four waves =
// 128 consecutive queries of one (utterance, head), whose windows together span at most 128 + context - 1 keys.  Each 32-key tile of that span is fetched
// and split into bf16 hi / lo ONCE per block, by all 256 threads, into LDS images laid out as the two products' A operands, and every wave whose window
// touches the tile multiplies from there.  Per wave and tile the kernel above issues 40 vector-memory instructions (32 of them 4-byte loads for V^T) and 40
// operand splits for 24 matrix instructions -- its matrix pipe is busy 17 % of the time (profiles/r4_pmc_mimi.txt); here a thread issues 10 loads and 8
// splits per tile, a wave 16 ds_read_b128.
//   K image (per half hi / lo): [32 keys][128 B = 64 dims bf16], the 16-byte chunk c (dims 8c .. 8c+7) of key k at position c ^ ((k >> 1) & 7): the
//     fragment read of step i (lane: key = lane & 31, g = lane >> 5 -> chunk 2i + g) is conflict-free for every 16 lanes.
//   V^T image (per half): [64 dims][64 B = 32 keys bf16 in the ORDER the score registers hold them: position 16 i + 8 g + 4 b + e = key 16 i + 8 b + 4 g + e],
//     chunk c = 2i + g of dim d at position c ^ aw_vswz(d) (conflict-free for the fragment reads AND for the staging stores: see aw_vswz).
// Two buffers: the loads of tile t + 1 are requested before tile t is multiplied, split and written behind it, one barrier per tile.
// Tiles are aligned to the block's first visible key (not each wave's own): the sums of a row are grouped differently from the kernel above for the first
// context + 96 positions of an utterance -- rounding-order differences, held by the same tolerances.
// ------------------------------------------------------------------------------------------------
constexpr int AW_KB = 32 * 128;   // bytes of one K half image
constexpr int AW_VB = 64 * 64;    // bytes of one V^T half image
constexpr int AW_TILE = 2 * AW_KB + 2 * AW_VB;   // 16 KB per tile: K hi | K lo | V^T hi | V^T lo

// chunk swizzle of the V^T image (64 rows of 64 B: four 16-byte chunks per row).  Round 4's (row >> 2) & 3 kept the fragment READS conflict-free but put the rows r and
// r + 2 that one 8-lane service group of the staging ds_write_b128 stores to on the same banks (a row is 64 B, the store banks repeat every 128 B): 2-way on
// every V^T store = the 5.87e6 conflict cycles on 6.12e6 LDS instructions of profiles/r4_pmc_mimi.txt (bank model: tools/probes/lds_conflicts.py's functions,
// 16 cycles per store instead of 8).  This one is conflict-free for both (model: 8 / 4): bit 0 = row bit 2, bit 1 = row bit 1 ^ row bit 3.
__device__ __forceinline__ int aw_vswz(int row) { return ((row >> 2) & 1) | ((((row >> 1) ^ (row >> 3)) & 1) << 1); }

template <int NW>   // waves = 32-query tiles per block (4: 128 queries, 12 key tiles of which a wave uses 9; 2: 64 queries, 10 of which it uses 9)
__global__ __launch_bounds__(64 * NW) void k_attn_window_lds(AttnArgs a, int qblocks) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * AW_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x;
    const int seg = blockIdx.y / qblocks, qb = blockIdx.y % qblocks;
    const int seg_rows = a.rows_per_seg;
    constexpr int QB = 32 * NW, SU = 4 / NW;         // queries per block; staging units per thread
    const int b0 = qb * QB;                          // the block's first query row of the segment
    const int r0 = b0 + wave * 32;                   // the wave's
    const bool wave_on = r0 < seg_rows;              // (a wave without queries still stages tiles and keeps the barriers' count)
    const int nq = wave_on ? min(32, seg_rows - r0) : 1;
    const int j = lane & 31, half = lane >> 5;
    const int my_q = min(j, nq - 1);
    const int ctx = a.context;
    const int pb_first = a.pos_base + b0, pb_last = a.pos_base + min(b0 + QB, seg_rows) - 1;   // positions of the block's queries
    const int p_first = a.pos_base + (wave_on ? r0 : b0), p_last = p_first + nq - 1;
    const int my_pos = p_first + my_q;
    const int jb_lo = max(0, pb_first - ctx + 1);    // first key any query of the block sees
    const int jw_lo = max(0, p_first - ctx + 1);     // ... this wave's
    const RowMap qm{a.q_ld, a.q_rows_per_batch, a.q_batch_stride}, om{a.out_ld, a.o_rows_per_batch, a.o_batch_stride};
    const int row = seg * seg_rows + (wave_on ? r0 : b0) + my_q;

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
    const float* kb = (const float*)a.k + (int64_t)seg * a.k_seg_stride + (int64_t)h * a.k_head_stride;
    const float* vb = (const float*)a.v + (int64_t)seg * a.k_seg_stride + (int64_t)h * a.k_head_stride;

    // ---- staging roles: 256 units over the block's 64 NW threads (unit id = tid + 64 NW u) ----
    // K: unit -> (key sk = id >> 3, dims 8 (id & 7) .. + 7): two float4, one 16-byte chunk of the image per half
    // V^T: unit -> (dim sd = id & 63, key group sg = id >> 6 (0..3): positions 8 sg .. 8 sg + 7 of the image row, i.e. chunk sg = 2i + g, keys 16 i + 8 b + 4 g + e)
    const int rs = (int)a.k_row_stride;             // (a segment's rows fit 32-bit element offsets: host check)
    float4 sk0[SU], sk1[SU];
    float sv[SU][8];
    auto stage_load = [&](int kt) {
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int id = tid + 64 * NW * u;
            const int sk = id >> 3, sc = id & 7, sd = id & 63, sg = id >> 6;
            const int v_key0 = 16 * (sg >> 1) + 4 * (sg & 1);   // keys v_key0 + e and v_key0 + 8 + e, e = 0..3
            const int key = min(kt + sk, pb_last);      // (keys past the block's last position are never visible: a valid row is read again)
            const float* kp = kb + sc * 8 + key * rs;
            sk0[u] = *reinterpret_cast<const float4*>(kp);
            sk1[u] = *reinterpret_cast<const float4*>(kp + 4);
            if (kt + 31 <= pb_last) {                   // (block-uniform) a whole tile: no clamping, one base + constant offsets
                const float* vp = vb + sd + (kt + v_key0) * rs;
#pragma unroll
                for (int e = 0; e < 8; e++) sv[u][e] = vp[((e & 3) + 8 * (e >> 2)) * rs];
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) sv[u][e] = vb[sd + min(kt + v_key0 + (e & 3) + 8 * (e >> 2), pb_last) * rs];
            }
        }
    };
    auto stage_store = [&](unsigned char* buf) {
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int id = tid + 64 * NW * u;
            const int sk = id >> 3, sc = id & 7, sd = id & 63, sg = id >> 6;
            const int k_dst = sk * 128 + ((sc ^ ((sk >> 1) & 7)) << 4);
            const int v_dst = sd * 64 + ((sg ^ aw_vswz(sd)) << 4);
            uint4 hq, lq;
            split2w(sk0[u].x, sk0[u].y, hq.x, lq.x);
            split2w(sk0[u].z, sk0[u].w, hq.y, lq.y);
            split2w(sk1[u].x, sk1[u].y, hq.z, lq.z);
            split2w(sk1[u].z, sk1[u].w, hq.w, lq.w);
            *reinterpret_cast<uint4*>(buf + k_dst) = hq;
            *reinterpret_cast<uint4*>(buf + AW_KB + k_dst) = lq;
            split2w(sv[u][0], sv[u][1], hq.x, lq.x);
            split2w(sv[u][2], sv[u][3], hq.y, lq.y);
            split2w(sv[u][4], sv[u][5], hq.z, lq.z);
            split2w(sv[u][6], sv[u][7], hq.w, lq.w);
            *reinterpret_cast<uint4*>(buf + 2 * AW_KB + v_dst) = hq;
            *reinterpret_cast<uint4*>(buf + 2 * AW_KB + AW_VB + v_dst) = lq;
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; r++) { o0[r] = 0.0f; o1[r] = 0.0f; }
    float m = -INFINITY, l = 0.0f;
    // fragment addresses of this lane inside a tile buffer
    const int ka = j * 128, ksw = (j >> 1) & 7;            // K: chunk 2 i + half
    const int va0 = j * 64, va1 = (j + 32) * 64, vsw0 = aw_vswz(j), vsw1 = aw_vswz(j + 32);   // V^T rows j and 32 + j: chunk 2 i + half

    stage_load(jb_lo);
    stage_store(lds);
    __syncthreads();
    int t = 0;
    for (int kt = jb_lo; kt <= pb_last; kt += 32, t ^= 1) {
        const unsigned char* buf = lds + t * AW_TILE;
        const bool more = kt + 32 <= pb_last;
        if (more) stage_load(kt + 32);
        if (wave_on && kt + 31 >= jw_lo && kt <= p_last) {   // the tile holds keys some query of this wave sees (wave-uniform)
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; r++) s[r] = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                FragW kh, kl;
                kh.q = *reinterpret_cast<const uint4*>(buf + ka + (((2 * i + half) ^ ksw) << 4));
                kl.q = *reinterpret_cast<const uint4*>(buf + AW_KB + ka + (((2 * i + half) ^ ksw) << 4));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh.v, qh[i].v, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl.v, qh[i].v, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh.v, ql[i].v, s, 0, 0, 0);
            }
            float tmax = -INFINITY;
            if (kt + 31 <= p_first && kt > p_last - ctx) {   // (wave-uniform) every key of the tile is visible to every query of the wave: no mask
#pragma unroll
                for (int r = 0; r < 16; r++) tmax = fmaxf(tmax, s[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int key = kt + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const bool ok = key <= my_pos && key > my_pos - ctx;
                    s[r] = ok ? s[r] : -INFINITY;
                    tmax = fmaxf(tmax, s[r]);
                }
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, WAVE));
            const float m_new = fmaxf(m, tmax);
            const float m_use = m_new == -INFINITY ? 0.0f : m_new;   // a tile may hold no visible key for this query yet
            const float alpha = __expf(m - m_use);
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
                for (int e = 0; e < 4; e++) split2w(s[8 * i + 2 * e], s[8 * i + 2 * e + 1], ph.u[e], pl.u[e]);
                const unsigned char* vt = buf + 2 * AW_KB;
                vh0.q = *reinterpret_cast<const uint4*>(vt + va0 + (((2 * i + half) ^ vsw0) << 4));
                vl0.q = *reinterpret_cast<const uint4*>(vt + AW_VB + va0 + (((2 * i + half) ^ vsw0) << 4));
                vh1.q = *reinterpret_cast<const uint4*>(vt + va1 + (((2 * i + half) ^ vsw1) << 4));
                vl1.q = *reinterpret_cast<const uint4*>(vt + AW_VB + va1 + (((2 * i + half) ^ vsw1) << 4));
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh0.v, ph.v, o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl0.v, ph.v, o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh0.v, pl.v, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh1.v, ph.v, o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl1.v, ph.v, o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh1.v, pl.v, o1, 0, 0, 0);
            }
        }
        if (more) stage_store(lds + (t ^ 1) * AW_TILE);
        __syncthreads();
    }
    l += __shfl_xor(l, 32, WAVE);
    if (wave_on && j < nq) {
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
        // (measured: Mimi phase of the benchmark batch 12.85 -> 12.50 ms with the shared tiles, 128 queries per block; 64 queries per block -- two waves, 10 tiles
        // of which a wave uses 9 instead of 12 / 9 -- 12.7 ms; profiles/r4_mimi_ab.txt)
        if ((int64_t)(a.pos_base + a.rows_per_seg) * a.k_row_stride < (1ll << 31)) {
            const int qblocks = (a.rows_per_seg + 127) / 128;
            hipLaunchKernelGGL(k_attn_window_lds<4>, dim3(a.heads, (unsigned)(segs * qblocks)), dim3(256), 0, stream, a, qblocks);
        } else hipLaunchKernelGGL((k_attn_window<false, false>), grid, dim3(256), 0, stream, a, qtiles);
    }
}

}  // namespace ptts
