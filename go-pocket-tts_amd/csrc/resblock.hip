// resblock.hip -- one SEANet residual block of the Mimi decoder as a single kernel (mimi.go:146-164,752-783):
//     uo = elu( u + conv_k1( elu( conv_k3( elu(u) ) + b1 ) ) + b2 )          (every reader of the sum applies ELU first)
// and, for the last block, the model's final causal convolution C -> 1 on top of it, writing PCM.
#include <cstdlib>

#include "kernels.h"
#include "device_util.h"

namespace ptts {

// As three GEMM launches the last two blocks (C = 128 and 64 channels at 3.84 M and 15.36 M rows for 64 x 10 s) moved
// u, the hidden tensor and the sum through HBM seven times over: ~26 GB per batch, the bulk of the decoder's traffic.
// Here a block owns TR consecutive rows of one utterance, reads them ONCE, keeps everything else on chip, and writes
// the sum (or 1/C-th of it: PCM) once:
//   A  u tile -> ELU -> bf16 hi/lo planes in LDS (XOR-swizzled 16-byte chunks), two rows of history in front
//   B  conv_k3 as an MFMA GEMM over the 3C-wide causal window: the window of row i is rows i-2..i of the planes, so
//      a k step is just a row offset.  +b1, ELU, split -> hidden planes in LDS
//   C  conv_k1 GEMM over the hidden planes; + b2 + u (re-read from L2), ELU -> global (uo), or f32 LDS tile (last block)
//   D  (last block) final conv: the sum goes back into the planes and a one-column MFMA product over the same windows
//      gives the PCM sample of every row
// Weights are read from fragment-ordered copies (model.cpp add_frag16), one contiguous 1-KiB burst per wave-instruction,
// L2-resident; products are computed transposed (weights as the MFMA row operand) so a lane ends up with four
// consecutive channels of one row.  Numerics are those of k_gemm3 (activations hi + lo, f32 weights hi + lo,
// f32 accumulation).  The first HALO rows of a tile only feed later rows and are recomputed by the neighbouring tile.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2r(float a, float b, unsigned& hi, unsigned& lo) {
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);
    f32x2 r = f - __builtin_convertvector(h, f32x2);
    bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

union FragR {
    bf16x8 v;
    uint4 q;
};

// PERS (bf16 weights): a block stays on its CU and walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...  Every weight
// fragment of the three products is copied into LDS once per block, so that the ONLY global fetches of a tile are its rows --
// and those are requested one tile ahead, right after the previous tile's rows have been consumed (vector loads return in
// order: with weight fragments fetched from L2 in between, a prefetch would only move the wait).  Without PERS a tile is a
// chain of four dependent fetches (rows, then each stage's fragments) behind three barriers and the waves sit parked 63 % of
// their cycles (profiles/r1_pmc_mimi_sq.txt).
template <int C, int H, int NW, bool FINAL, bool WBF16, bool PERS = false>
__global__ __launch_bounds__(NW * 64) void k_resblock(ResArgs a) {
    static_assert(!PERS || WBF16, "the persistent form keeps bf16 weight fragments in LDS");
    constexpr int TR = NW * 16;                       // rows per tile
    constexpr int HALO = FINAL ? 4 : 2;               // leading rows that are only inputs to later rows
    constexpr int TOUT = TR - HALO;
    constexpr int NTH = NW * 64;
    constexpr int ROWB = C * 2, HROWB = H * 2;        // bytes per plane row
    constexpr int CM = C / 8 - 1, HM = H / 8 - 1;     // chunk-swizzle masks (16-byte chunks per row - 1)
    constexpr int PLANE = (TR + 2) * ROWB, HPLANE = TR * HROWB;
    constexpr int EU_BYTES = 2 * PLANE;
    constexpr int F1 = (H / 16) * (3 * C / 32), F2 = (C / 16) * (H / 32), FF = FINAL ? 2 * (3 * C / 32) : 0;   // weight fragments (1 KiB each): conv k3, conv k1, final hi + lo
    constexpr int WL_BYTES = PERS ? (F1 + F2 + FF) * 1024 + (H + C + 4) * 4 : 0;   // + the three bias vectors
    __shared__ __attribute__((aligned(16))) unsigned char smem[EU_BYTES + 2 * HPLANE + WL_BYTES];
    unsigned char* eu_hi = smem;
    unsigned char* eu_lo = smem + PLANE;
    unsigned char* h_hi = smem + EU_BYTES;
    unsigned char* h_lo = h_hi + HPLANE;
    const uint4* wl1 = reinterpret_cast<const uint4*>(smem + EU_BYTES + 2 * HPLANE);   // PERS: the fragments in LDS
    const uint4* wl2 = wl1 + F1 * 64;
    const uint4* wlf = wl2 + F2 * 64;
    float* bl1 = reinterpret_cast<float*>(smem + EU_BYTES + 2 * HPLANE + (F1 + F2 + FF) * 1024);   // PERS: b1 [H], b2 [C], bf [1] (zeros when absent)
    float* bl2 = bl1 + H;
    float* blf = bl2 + C;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int tiles = (a.t1 - a.t0 + TOUT - 1) / TOUT;
    const int total = a.B * tiles;
    // The tile's rows are fetched in the distribution the MFMA RESULTS have (lane (r16, g) of wave w: tile row 16 w + r16, channels 16 j + 4 g .. + 3): what a lane
    // fetched is then exactly the residual operand of its stage-C results, and u is read from memory once.  (Round 4 fetched the tile in flat order -- 1-KB
    // wave-instructions -- and read the residual again behind stage A: "L2-hot" in intent, but the rows had been prefetched a whole tile earlier and half of the
    // second reads came from HBM: 2.98 GB fetched for a 1.97 GB input, profiles/r4_pmc_mimi.txt.)  A wave-instruction now covers 16 rows x 64 B.
    constexpr int V = C / 16;                         // float4 of the tile per thread (= TR * C / 4 / NTH)
    static_assert(V * NTH * 4 == TR * C, "a lane owns one tile row");
    float4 x[V];
    const int i_own = wave * 16 + r16;                // the tile row this lane fetches, splits, multiplies and stores
    auto request_rows = [&](int t) {                  // rows of tile t (clamped addresses; masked when consumed)
        const int bi_ = t / tiles, row0_ = a.t0 + (t % tiles) * TOUT - HALO;
        const float* ub_ = a.u + (int64_t)bi_ * a.u_bs + (int64_t)a.pad * C;
        const int gr = row0_ + i_own;
        const bool ok = gr >= -a.pad && gr < a.L;
        const float* rp = ub_ + (int64_t)(ok ? gr : 0) * C + 4 * g;
#pragma unroll
        for (int j = 0; j < V; j++) x[j] = *reinterpret_cast<const float4*>(rp + j * 16);
    };
    int tile = blockIdx.x;
    request_rows(tile);
    if constexpr (PERS) {
        uint4* dst = reinterpret_cast<uint4*>(smem + EU_BYTES + 2 * HPLANE);
        for (int i = tid; i < F1 * 64; i += NTH) dst[i] = reinterpret_cast<const uint4*>(a.w1)[i];
        for (int i = tid; i < F2 * 64; i += NTH) dst[F1 * 64 + i] = reinterpret_cast<const uint4*>(a.w2)[i];
        if constexpr (FINAL) {
            for (int i = tid; i < (FF / 2) * 64; i += NTH) {
                dst[(F1 + F2) * 64 + i] = reinterpret_cast<const uint4*>(a.wf_hi)[i];
                dst[(F1 + F2 + FF / 2) * 64 + i] = reinterpret_cast<const uint4*>(a.wf_lo)[i];
            }
        }
        // (a global load behind the prefetch would have to wait for it: the biases live in LDS too)
        if (tid < H) bl1[tid] = a.b1 ? a.b1[tid] : 0.0f;
        if (tid < C) bl2[tid] = a.b2 ? a.b2[tid] : 0.0f;
        if (FINAL && tid == 0) blf[0] = a.bf ? a.bf[0] : 0.0f;
    }
  for (;;) {
    const int bi = tile / tiles, tb = a.t0 + (tile % tiles) * TOUT;   // first output row of the tile
    const int row0 = tb - HALO;                                         // global row of tile row 0
    PcmRow pr{nullptr, 0, 0};                          // fetched here, where the wait for the tile's rows covers it
    if (FINAL && a.pcm_rows) pr = a.pcm_rows[bi];

    // ---- A: u -> elu -> hi/lo planes.  Rows before the utterance (beyond its zero history) or past its end are zeros ----
    {
        {
            const int gr = row0 + i_own;
            const bool ok = gr >= -a.pad && gr < a.L;
#pragma unroll
            for (int j = 0; j < V; j++) if (!ok) x[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < 2 * ROWB / 16) {                    // the two history rows in front of the tile only feed halo rows: zeros
            reinterpret_cast<uint4*>(eu_hi)[tid] = make_uint4(0, 0, 0, 0);
            reinterpret_cast<uint4*>(eu_lo)[tid] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < V; j++) {
            const int c = j * 16 + 4 * g;
            const int rho = i_own + 2;
            unsigned h01, l01, h23, l23;
            split2r(elu_fast(x[j].x), elu_fast(x[j].y), h01, l01);
            split2r(elu_fast(x[j].z), elu_fast(x[j].w), h23, l23);
            const int off = rho * ROWB + ((((c >> 3) ^ rho) & CM) << 4) + ((c & 4) << 1);
            *reinterpret_cast<uint2*>(eu_hi + off) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(eu_lo + off) = make_uint2(l01, l23);
        }
    }
    __syncthreads();

    const int i_lane = wave * 16 + r16;               // tile row this lane owns in the MFMA operands / results
    const int gr_lane = row0 + i_lane;
    const bool in_seq = gr_lane >= 0 && gr_lane < a.L;
    float4 ur[C / 16];                                // residual operand of stage C: the lane's own share of the tile (rows that are not in_seq are never stored)
#pragma unroll
    for (int n = 0; n < C / 16; n++) ur[n] = x[n];
    if constexpr (PERS) request_rows(min(tile + (int)gridDim.x, total - 1));   // past the block's last tile: a re-read that is never consumed
    // ---- B: hidden = elu(conv_k3(eu) + b1) ----
    {
        constexpr int NT = H / 16, KS = 3 * C / 32;
        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint4* w1 = (PERS ? wl1 : reinterpret_cast<const uint4*>(a.w1)) + lane;
        const uint4* w1l = reinterpret_cast<const uint4*>(a.w1_lo) + lane;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int tap = (s * 32) / C, c0 = (s * 32) % C;
            const int rho = i_lane + tap;             // window row i-2+tap, stored at rho = that + 2
            const int off = rho * ROWB + ((((c0 >> 3) + g) ^ rho) & CM) * 16;
            FragR xh, xl;
            xh.q = *reinterpret_cast<const uint4*>(eu_hi + off);
            xl.q = *reinterpret_cast<const uint4*>(eu_lo + off);
#pragma unroll
            for (int n = 0; n < NT; n++) {
                FragR wh;
                wh.q = w1[(n * KS + s) * 64];
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xh.v, acc[n], 0, 0, 0);
                PTTS_LO_MFMA(acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xl.v, acc[n], 0, 0, 0));
                if constexpr (!WBF16) {
                    FragR wl;
                    wl.q = w1l[(n * KS + s) * 64];
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl.v, xh.v, acc[n], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NT; n++) {                // lane: row i_lane, hidden channels n*16 + 4g .. +3
            const int ch = n * 16 + 4 * g;
            const float4 b = PERS ? *reinterpret_cast<const float4*>(bl1 + ch) : (a.b1 ? *reinterpret_cast<const float4*>(a.b1 + ch) : make_float4(0.f, 0.f, 0.f, 0.f));
            unsigned h01, l01, h23, l23;
            split2r(elu_fast(acc[n][0] + b.x), elu_fast(acc[n][1] + b.y), h01, l01);
            split2r(elu_fast(acc[n][2] + b.z), elu_fast(acc[n][3] + b.w), h23, l23);
            const int off = i_lane * HROWB + ((((ch >> 3) ^ i_lane) & HM) << 4) + ((ch & 4) << 1);
            *reinterpret_cast<uint2*>(h_hi + off) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(h_lo + off) = make_uint2(l01, l23);
        }
    }
    __syncthreads();

    // ---- C: sum = elu(u + conv_k1(hidden) + b2) ----
    {
        constexpr int NT = C / 16, KS = H / 32;
        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int gr = gr_lane;
        const uint4* w2 = (PERS ? wl2 : reinterpret_cast<const uint4*>(a.w2)) + lane;
        const uint4* w2l = reinterpret_cast<const uint4*>(a.w2_lo) + lane;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int off = i_lane * HROWB + (((s * 4 + g) ^ i_lane) & HM) * 16;
            FragR xh, xl;
            xh.q = *reinterpret_cast<const uint4*>(h_hi + off);
            xl.q = *reinterpret_cast<const uint4*>(h_lo + off);
#pragma unroll
            for (int n = 0; n < NT; n++) {
                FragR wh;
                wh.q = w2[(n * KS + s) * 64];
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xh.v, acc[n], 0, 0, 0);
                PTTS_LO_MFMA(acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xl.v, acc[n], 0, 0, 0));
                if constexpr (!WBF16) {
                    FragR wl;
                    wl.q = w2l[(n * KS + s) * 64];
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl.v, xh.v, acc[n], 0, 0, 0);
                }
            }
        }
        const bool store = !FINAL && i_lane >= HALO && gr < a.t1 && in_seq;
        float* orow = FINAL ? nullptr : a.uo + (int64_t)bi * a.u_bs + (int64_t)(a.pad + (in_seq ? gr : 0)) * C;
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const int ch = n * 16 + 4 * g;
            const float4 b = PERS ? *reinterpret_cast<const float4*>(bl2 + ch) : (a.b2 ? *reinterpret_cast<const float4*>(a.b2 + ch) : make_float4(0.f, 0.f, 0.f, 0.f));
            float4 v;
            v.x = elu_fast(ur[n].x + (acc[n][0] + b.x)); v.y = elu_fast(ur[n].y + (acc[n][1] + b.y));
            v.z = elu_fast(ur[n].z + (acc[n][2] + b.z)); v.w = elu_fast(ur[n].w + (acc[n][3] + b.w));
            if constexpr (FINAL) {                    // the sum replaces elu(u) in the planes (dead since stage B's barrier);
                if (!in_seq) v = make_float4(0.f, 0.f, 0.f, 0.f);   // rows before the utterance are the final conv's zero padding
                const int rho = i_lane + 2;
                unsigned h01, l01, h23, l23;
                split2r(v.x, v.y, h01, l01);
                split2r(v.z, v.w, h23, l23);
                const int off = rho * ROWB + ((((ch >> 3) ^ rho) & CM) << 4) + ((ch & 4) << 1);
                *reinterpret_cast<uint2*>(eu_hi + off) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(eu_lo + off) = make_uint2(l01, l23);
            } else if (store) {
                *reinterpret_cast<float4*>(orow + ch) = v;
            }
        }
    }
    if constexpr (FINAL) {
        // ---- D: pcm[row] = bf + sum_{tap, c} sum[row-2+tap][c] * wf[tap*C + c]: the same windowed product as stage B with a
        // one-column weight matrix (column 0 of a 16-column fragment; f32 weights, so hi and lo planes) ----
        __syncthreads();
        constexpr int KS = 3 * C / 32;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint4* wfh = (PERS ? wlf : reinterpret_cast<const uint4*>(a.wf_hi)) + lane;
        const uint4* wfl = (PERS ? wlf + (FF / 2) * 64 : reinterpret_cast<const uint4*>(a.wf_lo)) + lane;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int tap = (s * 32) / C, c0 = (s * 32) % C;
            const int rho = i_lane + tap;
            const int off = rho * ROWB + ((((c0 >> 3) + g) ^ rho) & CM) * 16;
            FragR xh, xl, wh, wl;
            xh.q = *reinterpret_cast<const uint4*>(eu_hi + off);
            xl.q = *reinterpret_cast<const uint4*>(eu_lo + off);
            wh.q = wfh[s * 64];
            wl.q = wfl[s * 64];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xh.v, acc, 0, 0, 0);
            PTTS_LO_MFMA(acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xl.v, acc, 0, 0, 0));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl.v, xh.v, acc, 0, 0, 0);
        }
        const int gr = row0 + i_lane;                 // result column 0 sits in register 0 of lane group 0
        const float smp = acc[0] + (PERS ? blf[0] : (a.bf ? a.bf[0] : 0.0f));
        if (a.pcm_rows) {
            // the tile's TOUT samples are gathered in LDS (the hidden planes are free by now) and leave as 16-byte (f32) or
            // 8-byte (int16) pieces per lane, contiguous over the first lanes of the block: sized for a PCIe write
            float* stage = reinterpret_cast<float*>(h_hi);
            if (g == 0) stage[i_lane] = smp;
            __syncthreads();
            const int lim = min(min(a.t1, a.L), pr.lim);
            const int j = tid * 4, idx = tb + j;
            if (j < TOUT && idx < lim) {
                const float4 v = *reinterpret_cast<const float4*>(stage + HALO + j);
                if (pr.s16) {
                    int16_t* dst = reinterpret_cast<int16_t*>(pr.dst) + idx;
                    const int s0 = pcm16_one(v.x), s1 = pcm16_one(v.y), s2 = pcm16_one(v.z), s3 = pcm16_one(v.w);
                    if (idx + 3 < lim) *reinterpret_cast<uint2*>(dst) = make_uint2((unsigned)(s0 & 0xffff) | ((unsigned)s1 << 16), (unsigned)(s2 & 0xffff) | ((unsigned)s3 << 16));
                    else {
                        dst[0] = (int16_t)s0;
                        if (idx + 1 < lim) dst[1] = (int16_t)s1;
                        if (idx + 2 < lim) dst[2] = (int16_t)s2;
                    }
                } else {
                    float* dst = reinterpret_cast<float*>(pr.dst) + idx;
                    if (idx + 3 < lim) *reinterpret_cast<float4*>(dst) = v;
                    else {
                        dst[0] = v.x;
                        if (idx + 1 < lim) dst[1] = v.y;
                        if (idx + 2 < lim) dst[2] = v.z;
                    }
                }
            }
        } else if (g == 0 && i_lane >= HALO && gr < a.t1 && gr < a.L) {
            a.pcm[(int64_t)bi * a.pcm_bs + gr] = smp;
        }
    }
    if constexpr (!PERS) break;
    tile += gridDim.x;
    if (tile >= total) break;
    __syncthreads();   // the planes and the staging rows are rewritten by the next tile
  }
}

bool resblock_supported(const ResArgs& a) {
    const bool dims = (a.C == 64 && a.H == 32) || (a.C == 128 && a.H == 64);
    return dims && a.k1 == 3 && a.k2 == 1 && a.w1 && a.w2 && (a.w_bf16 || (a.w1_lo && a.w2_lo)) && a.pad >= 2 && aligned16(a.u) &&
           (a.final_conv ? (a.kf == 3 && a.wf_hi && a.wf_lo && (a.pcm_rows ? a.t0 % 4 == 0 : a.pcm != nullptr)) : (a.uo != nullptr && aligned16(a.uo))) && a.u_bs % 4 == 0 && a.t1 > a.t0;
}

template <int C, int H, int NW>
static void launch_rb(const ResArgs& a, hipStream_t stream) {
    const int tout = NW * 16 - (a.final_conv ? 4 : 2);
    const int tiles = (a.t1 - a.t0 + tout - 1) / tout;
    dim3 grid((unsigned)(a.B * tiles));
    if (a.final_conv) {
        if constexpr (C == 64) {
            static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
            const int resident = 2 * cus;                    // 77 KB of LDS and 116 registers per lane: two blocks per CU
            if (a.w_bf16 && a.B * tiles >= resident * 8) {   // enough tiles per block to amortise its 28-KB weight copy
                hipLaunchKernelGGL((k_resblock<C, H, NW, true, true, true>), dim3((unsigned)resident), dim3(NW * 64), 0, stream, a);
                return;
            }
        }
        if (a.w_bf16) hipLaunchKernelGGL((k_resblock<C, H, NW, true, true>), grid, dim3(NW * 64), 0, stream, a);
        else hipLaunchKernelGGL((k_resblock<C, H, NW, true, false>), grid, dim3(NW * 64), 0, stream, a);
    } else {
        if constexpr (C == 128) {
            if (a.w_bf16) {   // persistent as well: 1458 -> 1055 us at batch 64
                constexpr int PNW = 6;                                  // 96-row tiles: 139 KB of LDS with the 64 KB of weights, one block per CU
                static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
                const int ptiles = (a.t1 - a.t0 + PNW * 16 - 2 - 1) / (PNW * 16 - 2);
                if (a.B * ptiles >= cus * 8) {
                    hipLaunchKernelGGL((k_resblock<C, H, PNW, false, true, true>), dim3((unsigned)cus), dim3(PNW * 64), 0, stream, a);
                    return;
                }
            }
        }
        if (a.w_bf16) hipLaunchKernelGGL((k_resblock<C, H, NW, false, true>), grid, dim3(NW * 64), 0, stream, a);
        else hipLaunchKernelGGL((k_resblock<C, H, NW, false, false>), grid, dim3(NW * 64), 0, stream, a);
    }
}

void launch_resblock(const ResArgs& a, hipStream_t stream) {
    note_launch(a.final_conv ? "k_resblock+final" : "k_resblock");
    if (a.C == 64) launch_rb<64, 32, 8>(a, stream);    // 128-row tiles, ~50 KB of LDS: three blocks per CU
    else launch_rb<128, 64, 4>(a, stream);             // 64-row tiles, same footprint
}

}  // namespace ptts
