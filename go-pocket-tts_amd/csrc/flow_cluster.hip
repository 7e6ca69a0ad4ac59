// flow_cluster.hip -- the residual blocks of the flow net as one launch (the AR step's twelve 512 x 512 linears).
#include <hip/hip_ext.h>

#include "../../include/ptts.h"
#include "common.h"
#include "kernels.h"
#include "device_util.h"

namespace ptts {

// ------------------------------------------------------------------------------------------------
// flowResBlock.Forward x depth (flow_net.go:116-172):  h = silu(mlp0(modulate(LayerNorm(x), shift, scale)));  x += gate * mlp2(h)
//
// As launches (runtime.cpp step_core) these are 2 x depth dependent k_skinny launches of 4-5.6 us each for 0.5 MB of weights apiece: launch
// boundary, a cold argument block, a cold weight round trip and the store drain every time.  Here the chain stays inside one launch:
//   * a 12-row tile of the batch belongs to EIGHT workgroups; workgroup cb owns output columns [64 cb, 64 cb + 64) of every linear -- its weight
//     fragments are requested one linear ahead (they depend on nothing) and wait in registers;
//   * after each linear the eight exchange their 16 x 64 pieces so that each holds whole rows again (LayerNorm and the next product need them).
//     The exchange is the guide's tagged granule: every value travels as one aligned 8-byte {value, tag} written by a single write-through (sc1)
//     store, the reader sweeps its row with sc1 loads until all tags match -- the data is its own flag: no fence, no drain, no barrier on either
//     side (cdna_hip_programming.md Guideline 16, R2; a row is 4 KB of granules);
//   * tags count the exchanges of the tile across launches (a launch takes its base from a word in device memory that workgroup 0 of the tile
//     advances at its end: every peer has read it by then, or it could not have published what workgroup 0 consumed last), so a granule left by an
//     earlier exchange -- or an earlier launch, or an earlier replay of the same graph -- can never be taken for the awaited one; two buffers in
//     turn (h / x), since a workgroup may publish exchange e + 1 while a peer still sweeps e;
//   * every sweep is bounded; a sweep that gives up raises the fault word and the launch runs to its end on whatever it has (the host checks the word
//     with the step counters and fails the batch).
// The arithmetic is k_skinny's, instruction for instruction where it matters (the operand split, the k permutation of the fragment-ordered weights,
// the order of the K-quarter sums, LayerNorm, epilogues), with the two MFMA operands exchanged so that a lane ends up with four consecutive columns
// of one row (two 16-byte granule stores).
// Work split inside a workgroup: see the kernel.
// ------------------------------------------------------------------------------------------------
typedef __bf16 fc_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fc_bf16x2 __attribute__((ext_vector_type(2)));
typedef float fc_f32x2 __attribute__((ext_vector_type(2)));
typedef float fc_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned fc_u32x4 __attribute__((ext_vector_type(4)));

constexpr int FC_C = 512;                   // width of the flow net (host: flow_cluster_supported)
constexpr int FC_RB = 2048, FC_CMASK = 127;   // LDS image: bytes and 16-byte chunks (- 1) per row (bf16 x 1024: k_skinny's NJ <= 4 image)
constexpr unsigned FC_SPIN_LIMIT = 1u << 15;  // sweeps of a row before giving up (a pass is >= 0.5 us: >= 16 ms)
constexpr int FC_FAULT_WORD = 32 * kFlowClusterMaxTiles;   // behind the tiles' tag-base words (32 words apart)

union FcFrag {
    fc_bf16x8 v;
    uint4 q;
};

__device__ __forceinline__ void fc_split2(float a, float b, unsigned& hi, unsigned& lo) {   // (skinny.hip split2)
    fc_f32x2 f = {a, b};
    fc_bf16x2 h = __builtin_convertvector(f, fc_bf16x2);
    fc_f32x2 r = f - __builtin_convertvector(h, fc_f32x2);
    fc_bf16x2 l = __builtin_convertvector(r, fc_bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

// wave `wave`'s row of the LDS image from the lane's 2 x 4 values (columns 4 lane .. and 4 (lane + 64) ..)
__device__ __forceinline__ void fc_stage(unsigned char* Xh, unsigned char* Xl, int wave, int lane, const float4 (&xr)[2]) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int k = (lane + 64 * j) * 4;
        unsigned h01, l01, h23, l23;
        fc_split2(xr[j].x, xr[j].y, h01, l01);
        fc_split2(xr[j].z, xr[j].w, h23, l23);
        const int off = wave * FC_RB + ((((k >> 3) ^ wave) & FC_CMASK) << 4) + ((k & 4) << 1);
        *reinterpret_cast<uint2*>(&Xh[off]) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(&Xl[off]) = make_uint2(l01, l23);
    }
}

// the lane's 8 values of its wave's row from the granule buffer, once every tag is `tag`.
// A wave WATCHES two granule pairs of every producing wave (the ends of its second and fourth lane group's stores: 1 KB per wave and pass) and reads its
// row when those have turned -- then checks every tag of what it read (the pairs it watched say nothing certain about their neighbours) and goes back to
// watching if one is old.  Less traffic beats lower latency here, measured on one box in rotation (AR loop of the benchmark batch, ms): re-reading the whole
// row every pass 36.0; the row requested together with the watch granules from the third pass on (one round trip less when they have turned) 36.6-36.8
// against 35.9 without; 0.2 us between passes instead of 0.03: 35.8 against 36.0.
template <typename RS>
__device__ __forceinline__ bool fc_sweep(RS rs, int row_off, int lane, unsigned tag, float4 (&xr)[2]) {
    fc_u32x4 g0 = {0u, 0u, 0u, 0u}, g1 = g0, g2 = g0, g3 = g0;
    bool got = true;
    const int voff = row_off + lane * 32;
    const int watch = row_off + ((lane >> 1) * 16 + 6 + 8 * (lane & 1)) * 8;   // producing wave lane >> 1 (16 columns each): its columns 6, 7 / 14, 15
    for (unsigned spins = 0;; spins++) {
        asm volatile("" ::: "memory");   // (the loads are re-issued every pass)
        const fc_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, watch, 0, 16);   // aux 16: sc1
        if (__all(w[1] == tag && w[3] == tag)) {
            asm volatile("" ::: "memory");
            g0 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 16);
            g1 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16, 0, 16);
            g2 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 2048, 0, 16);
            g3 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 2064, 0, 16);
            const bool ok = g0[1] == tag && g0[3] == tag && g1[1] == tag && g1[3] == tag && g2[1] == tag && g2[3] == tag && g3[1] == tag && g3[3] == tag;
            if (__all(ok)) break;
        }
        if (spins > FC_SPIN_LIMIT) { got = false; g0 = g1 = g2 = g3 = fc_u32x4{0u, 0u, 0u, 0u}; break; }
        __builtin_amdgcn_s_sleep(8);   // ~0.2 us between passes: watching harder is slower (below)
    }
    xr[0] = make_float4(__uint_as_float(g0[0]), __uint_as_float(g0[2]), __uint_as_float(g1[0]), __uint_as_float(g1[2]));
    xr[1] = make_float4(__uint_as_float(g2[0]), __uint_as_float(g2[2]), __uint_as_float(g3[0]), __uint_as_float(g3[2]));
    return got;
}

template <typename RS>
__device__ __forceinline__ void fc_publish(RS rs, int voff, unsigned tag, const float (&v)[4]) {
    const fc_u32x4 a = {__float_as_uint(v[0]), tag, __float_as_uint(v[1]), tag}, b = {__float_as_uint(v[2]), tag, __float_as_uint(v[3]), tag};
    __builtin_amdgcn_raw_buffer_store_b128(a, rs, voff, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(b, rs, voff + 16, 0, 16);
}

// The launch's two roles (wave-uniform branch): waves 0..3 multiply and publish -- wave cg owns 16 of the workgroup's 64 columns over the WHOLE K (its
// weight fragments, 16 KB, sit in registers and are re-requested for the next linear as soon as the products have read them) --, waves 4..15 each stage
// one of the tile's TWELVE rows: sweep it out of the granule buffer, LayerNorm / modulate it, split it into the LDS image.  Why not k_skinny's split,
// where every wave stages a row and four of them also store (the first cut): a wave's memory operations complete in issue order, so a wave that has
// just published cannot take delivery of a single granule of the next exchange before its own write-through stores have been acknowledged by
// memory -- in-kernel stamps put a hop at 3.0-3.5 us for the block while the waves that had stored nothing held their rows after 1.1-1.5.  With the
// roles apart nobody waits for an acknowledgement (the multiplying waves next touch memory when they need the following linear's weights, a hop later),
// the stagers sweep while the products run, and one barrier per linear ("image complete") is the only meeting point: the image of linear p + 1 cannot
// be written before this workgroup's own multiplying waves have published linear p -- by which time they have read image p.
constexpr int FC_ROWS = kFlowClusterRows;          // rows per tile: one per staging wave
constexpr int FC_THREADS = 64 * (4 + FC_ROWS);
constexpr int FC_TILE_GRANULES = 2 * 16 * FC_C;   // granules of a tile's two buffers (16-row pitch)

template <bool STAMP>
__global__ __launch_bounds__(FC_THREADS) void k_flow_cluster(FlowClusterArgs a) {
    // (measurement build, PTTS_FC_STAMPS: 100-MHz timestamps of one multiplying and one staging wave of every workgroup at the phase boundaries)
#define FC_STAMP(i) do { if (STAMP && lane == 0) a.stamps[blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    __shared__ __attribute__((aligned(16))) unsigned char Xh[16 * FC_RB];   // (rows 12..15 are never written: they feed output rows nobody stores)
    __shared__ __attribute__((aligned(16))) unsigned char Xl[16 * FC_RB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroups that stream the same weight columns are 8-congruent in dispatch order (one XCD, one L2 copy of the weights: speed only)
    const int cb = blockIdx.x & 7, tile = blockIdx.x >> 3;
    const int m0 = tile * FC_ROWS;
    unsigned* const sync = a.sync + tile * 32;
    const unsigned base = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    auto rs = __builtin_amdgcn_make_buffer_rsrc(a.xbuf + (size_t)tile * FC_TILE_GRANULES, 0, FC_TILE_GRANULES * 8, 0x00020000);
    constexpr int BUF1 = 16 * FC_C * 8;   // second buffer (the h exchanges)

    if (wave < 4) {
        // ================= multiplying waves =================
        const int cg = wave, q = lane >> 4, i16 = lane & 15;
        const bool s_ok = i16 < FC_ROWS && m0 + i16 < a.rows;   // this lane's row exists
        const int64_t srow = s_ok ? m0 + i16 : 0;
        const int scol = cb * 64 + cg * 16 + q * 4;
        const int pub_off = i16 * (FC_C * 8) + scol * 8;        // the lane's four granules
        // the wave's weight fragments: tile of 16 columns, all four super-steps of the fragment-ordered copy (model.cpp add_tiled: [tile][ss][4][64 lanes] x 16 B)
        const int64_t wfrag = ((int64_t)(cb * 4 + cg) * 16) * 64 + lane;
        uint4 w[16];
        {
            const uint4* s0 = reinterpret_cast<const uint4*>(a.w0[0]) + wfrag;
#pragma unroll
            for (int i = 0; i < 16; i++) w[i] = s0[i * 64];
        }
        float4 res = *reinterpret_cast<const float4*>(a.fx_in + srow * FC_C + scol);   // the lane's piece of the residual stream, in registers across the blocks
        // 16 columns x 16 rows over K = 512: per K quarter an (hi, lo) accumulator pair, the quarters added in k_skinny's order
        auto product = [&](float (&acc)[4]) {
#pragma unroll
            for (int ss = 0; ss < 4; ss++) {
                fc_f32x4 acc_h = {0.f, 0.f, 0.f, 0.f}, acc_l = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    // chunk c = 16 ss + 4 q + s of row i16 sits at c ^ i16 (fc_stage): the quarter is an immediate offset of the four per-s addresses
                    const int off = i16 * FC_RB + (((q * 4 + s) ^ i16) << 4) + ss * 256;
                    FcFrag xh, xl, wv;
                    xh.q = *reinterpret_cast<const uint4*>(&Xh[off]);
                    xl.q = *reinterpret_cast<const uint4*>(&Xl[off]);
                    wv.q = w[ss * 4 + s];
                    acc_h = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv.v, xh.v, acc_h, 0, 0, 0);
                    acc_l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv.v, xl.v, acc_l, 0, 0, 0);
                }
                const fc_f32x4 accv = acc_h + acc_l;
                if (ss == 0) { acc[0] = accv[0]; acc[1] = accv[1]; acc[2] = accv[2]; acc[3] = accv[3]; }
                else { acc[0] += accv[0]; acc[1] += accv[1]; acc[2] += accv[2]; acc[3] += accv[3]; }
            }
        };
        // The following linear's fragments are requested right behind the products that read the current ones -- IN FRONT of the publish: behind it they would
        // not even be issued before the write-through stores are on their way (1.3 us, stamps), and a hop later is when they are needed.  (Tried: in the
        // gaps between the quarters' MFMAs -- spills at 128 registers; with 8-row tiles and 12 waves to make room, the kernel was 5 us slower.)
        auto next_weights = [&](const void* wt) {
            const uint4* sw = reinterpret_cast<const uint4*>(wt) + wfrag;
#pragma unroll
            for (int i = 0; i < 16; i++) w[i] = sw[i * 64];
        };
        if (wave == 0) FC_STAMP(0);
#pragma unroll 1
        for (int r = 0; r < a.depth; r++) {
            const int rn = min(r + 1, a.depth - 1);
            // this block's pointers out of the argument block NOW, in one batch of scalar loads: fetched where they are first used, each is a cold scalar-cache
            // round trip (~1 us) on the critical path -- between the publish and the weight requests behind it the first cut of this order lost 1.3 us to one
            const void* const pw2 = a.w2[r]; const void* const pw0n = a.w0[rn];
            const float* const pb0 = a.b0[r]; const float* const pb2 = a.b2[r];
            asm volatile("" ::"s"(pw2), "s"(pw0n), "s"(pb0), "s"(pb2));
            const float4 bias0 = *reinterpret_cast<const float4*>(pb0 + scol);
            float acc[4];
            // ---- mlp0: h = silu(W0 y + b0) ----
            __syncthreads();   // image complete
            if (wave == 0) FC_STAMP(1 + 4 * r);
            product(acc);
            next_weights(pw2);
            if (STAMP && wave == 0 && r == 2) { if (acc[0] == 1.2345e-30f) FC_STAMP(60); FC_STAMP(56); }   // (the comparison makes the stamp wait for the sums)
            {
                const float h[4] = {silu1(acc[0] + bias0.x), silu1(acc[1] + bias0.y), silu1(acc[2] + bias0.z), silu1(acc[3] + bias0.w)};
                if (STAMP && wave == 0 && r == 2) { if (h[0] == 1.2345e-30f) FC_STAMP(60); FC_STAMP(58); }
                if (s_ok && !(a.inject && tile == 0 && cb == 7 && r == a.inject - 1)) fc_publish(rs, BUF1 + pub_off, base + 2 * r + 1, h);   // (inject: ptts_debug_flow_cluster_inject)
            }
            if (STAMP && wave == 0 && r == 2) FC_STAMP(57);
            if (wave == 0) FC_STAMP(2 + 4 * r);
            // ---- mlp2: x += gate * (W2 h + b2) ----
            const float4 bias2 = *reinterpret_cast<const float4*>(pb2 + scol);   // (requested a hop ahead of their use)
            const float4 gate = *reinterpret_cast<const float4*>(a.ada + srow * a.ldmod + (int64_t)(r * 3 + 2) * FC_C + scol);
            __syncthreads();
            if (wave == 0) FC_STAMP(3 + 4 * r);
            product(acc);
            next_weights(pw0n);   // (the last block re-reads its own)
            res.x = res.x + gate.x * (acc[0] + bias2.x);
            res.y = res.y + gate.y * (acc[1] + bias2.y);
            res.z = res.z + gate.z * (acc[2] + bias2.z);
            res.w = res.w + gate.w * (acc[3] + bias2.w);
            if (s_ok) {
                if (r + 1 < a.depth) {
                    const float v[4] = {res.x, res.y, res.z, res.w};
                    fc_publish(rs, pub_off, base + 2 * r + 2, v);
                } else *reinterpret_cast<float4*>(a.fx_out + srow * FC_C + scol) = res;
            }
            if (wave == 0) FC_STAMP(4 + 4 * r);
        }
        if (cb == 0 && tid == 0) __hip_atomic_store(sync, base + 2u * (unsigned)a.depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == 0) FC_STAMP(31);
    } else {
        // ================= staging waves: row (wave - 4) of the tile =================
        const int row = wave - 4;
        const bool row_ok = m0 + row < a.rows;      // (wave-uniform; a wave without a row only keeps the barriers' count)
        const int64_t mrow = row_ok ? m0 + row : 0;
        const int kc0 = lane * 4, kc1 = (lane + 64) * 4;
        const int sweep_off = row * (FC_C * 8);
        unsigned fault = 0;
#pragma unroll 1
        for (int r = 0; r < a.depth; r++) {
            if (row_ok) {
                // ---- the residual stream's row: adaLN prologue of mlp0 ----
                float4 xr[2];
                const float* lnw = a.ln_w[r]; const float* lnb = a.ln_b[r];
                const float* shift = a.ada + mrow * a.ldmod + (int64_t)(r * 3) * FC_C; const float* scale = shift + FC_C;
                const float4 lw0 = *reinterpret_cast<const float4*>(lnw + kc0), lw1 = *reinterpret_cast<const float4*>(lnw + kc1);
                const float4 lb0 = *reinterpret_cast<const float4*>(lnb + kc0), lb1 = *reinterpret_cast<const float4*>(lnb + kc1);
                const float4 lc0 = *reinterpret_cast<const float4*>(scale + kc0), lc1 = *reinterpret_cast<const float4*>(scale + kc1);
                const float4 lh0 = *reinterpret_cast<const float4*>(shift + kc0), lh1 = *reinterpret_cast<const float4*>(shift + kc1);
                if (r == 0) {
                    xr[0] = *reinterpret_cast<const float4*>(a.fx_in + mrow * FC_C + kc0);
                    xr[1] = *reinterpret_cast<const float4*>(a.fx_in + mrow * FC_C + kc1);
                } else if (!fc_sweep(rs, sweep_off, lane, base + 2 * r, xr)) fault = 1;
                if (wave == 4) FC_STAMP(32 + 4 * r);
                // LayerNorm over the row, biased variance (linear.go:295-309) -- k_skinny's prologue (PRO_LN | PRO_AFFINE | PRO_MOD)
                fc_f32x2 s2 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 2; j++) s2 += fc_f32x2{xr[j].x, xr[j].z} + fc_f32x2{xr[j].y, xr[j].w};
                const float rk = __builtin_amdgcn_rcpf((float)FC_C);
                const float mean = wave_sum_dpp(s2.x + s2.y) * rk;
                const fc_f32x2 m2 = {mean, mean};
                fc_f32x2 v2 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const fc_f32x2 da = fc_f32x2{xr[j].x, xr[j].y} - m2, db = fc_f32x2{xr[j].z, xr[j].w} - m2;
                    v2 += da * da + db * db;
                }
                const float inv_std = __builtin_amdgcn_rsqf(wave_sum_dpp(v2.x + v2.y) * rk + a.eps[r]);
                const fc_f32x2 is2 = {inv_std, inv_std};
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float4 lw = j ? lw1 : lw0, lb = j ? lb1 : lb0, lc = j ? lc1 : lc0, lh = j ? lh1 : lh0;
                    fc_f32x2 oa = (fc_f32x2{xr[j].x, xr[j].y} - m2) * is2, ob = (fc_f32x2{xr[j].z, xr[j].w} - m2) * is2;
                    oa = oa * fc_f32x2{lw.x, lw.y} + fc_f32x2{lb.x, lb.y};
                    ob = ob * fc_f32x2{lw.z, lw.w} + fc_f32x2{lb.z, lb.w};
                    const fc_f32x2 one = {1.0f, 1.0f};
                    oa = oa * (fc_f32x2{lc.x, lc.y} + one) + fc_f32x2{lh.x, lh.y};
                    ob = ob * (fc_f32x2{lc.z, lc.w} + one) + fc_f32x2{lh.z, lh.w};
                    xr[j] = make_float4(oa.x, oa.y, ob.x, ob.y);
                }
                fc_stage(Xh, Xl, row, lane, xr);
            }
            if (wave == 4) FC_STAMP(33 + 4 * r);
            __syncthreads();   // image complete (mlp0)
            if (row_ok) {
                // ---- h's row: as it is ----
                float4 xr[2];
                if (!fc_sweep(rs, BUF1 + sweep_off, lane, base + 2 * r + 1, xr)) fault = 1;
                if (wave == 4) FC_STAMP(34 + 4 * r);
                fc_stage(Xh, Xl, row, lane, xr);
            }
            if (wave == 4) FC_STAMP(35 + 4 * r);
            __syncthreads();   // image complete (mlp2)
        }
        if (fault && lane == 0) atomicOr(a.sync + FC_FAULT_WORD, 1u);
    }
#undef FC_STAMP
}

bool flow_cluster_supported(const FlowClusterArgs& a, int C) {
    if (C != FC_C || a.rows <= 0 || a.rows > kFlowClusterMaxTiles * kFlowClusterRows || a.depth <= 0 || a.depth > FC_MAX_DEPTH || a.ldmod % 4 != 0 || !a.xbuf || !a.sync) return false;
    if (!aligned16(a.fx_in) || !aligned16(a.fx_out) || !aligned16(a.ada)) return false;
    for (int r = 0; r < a.depth; r++)
        if (!a.w0[r] || !a.w2[r] || !a.b0[r] || !a.b2[r] || !a.ln_w[r] || !a.ln_b[r] || !aligned16(a.b0[r]) || !aligned16(a.b2[r]) || !aligned16(a.ln_w[r]) || !aligned16(a.ln_b[r])) return false;
    return true;
}

bool flow_cluster_fits(int rows, int device) {
    // A tile's eight workgroups hand rows to each other inside the launch and spin (bounded) on each other's granules: the whole grid must be resident at
    // once.  A workgroup is 16 waves and 64 KB of LDS -- one per CU on gfx950 -- so the grid must not exceed what the device holds of them.
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) return false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_flow_cluster<false>, FC_THREADS, 0) != hipSuccess || per_cu <= 0) { (void)hipGetLastError(); return false; }
    return 8 * ((rows + FC_ROWS - 1) / FC_ROWS) <= cus * per_cu;
}

void launch_flow_cluster(const FlowClusterArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    note_launch("k_flow_cluster");
    const dim3 grid(8 * ((a.rows + FC_ROWS - 1) / FC_ROWS));
    if (a.stamps) hipLaunchKernelGGL(k_flow_cluster<true>, grid, dim3(FC_THREADS), 0, stream, a);
    else if (ev0) hipExtLaunchKernelGGL(k_flow_cluster<false>, grid, dim3(FC_THREADS), 0, stream, ev0, ev1, 0, a);
    else hipLaunchKernelGGL(k_flow_cluster<false>, grid, dim3(FC_THREADS), 0, stream, a);
}

}  // namespace ptts
