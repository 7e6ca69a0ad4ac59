// resblock_up.hip -- the last transposed convolution of the SEANet decoder, its residual block and the model's final convolution as
// ONE kernel (mimi.go:740-788: decoder.model.8 convtr 128 -> 64 stride 4, model.9 block, ELU, model.11 conv 64 -> 1):
//     u   = convtr( x )                         x: the previous block's output (already behind its ELU), 128 channels at 6 kHz
//     s   = elu( u + conv_k1( elu( conv_k3( elu(u) ) + b1 ) ) + b2 )
//     pcm = conv_k3( s ) + bf
// As two launches (k_gemm_wres<256,256>, k_resblock<64,32,...,FINAL>) u crosses HBM twice: 3.93 GB written and 3.93 GB read back per
// batch of 64 x 10 s, 1.75 ms of the decoder's 7.7 ms of HBM time (the decoder's time is its matrix time plus
// its HBM time).  Here a block owns 128 consecutive output rows (= 32 input rows) of one utterance, computes their u in registers
// from 33 input rows and hands it to k_resblock's stages through LDS; u never exists in memory.
//
//   X  x rows (33 x 128 f32, requested one tile ahead) -> bf16 hi / lo planes in LDS (XOR-swizzled 16-byte chunks)
//   0  u = transposed convolution as a product [32 rows x 256 k] x [256 k x 256 n], k = (x[t-1] | x[t]), n = (phase, channel).  The
//      256 x 256 bf16 weights do not fit LDS beside the rest: every wave keeps ITS 32 columns of them in registers for the whole kernel
//      (64 VGPRs: two 16-column tiles x 8 k steps, fragment-ordered; a tile = all four phases of four channels, so that a wave's lanes end
//      up with 64 consecutive output rows: conflict-free LDS writes) and multiplies them with all 32 rows.  u (+ bias) goes to LDS twice:
//      as f32 (the residual operand of stage C) and, behind ELU, as the hi / lo planes stage B reads -- what k_resblock's stage A writes
//   B, C, D  as in k_resblock (resblock.hip): conv_k3 + ELU -> hidden planes, conv_k1 + residual + ELU -> planes, final conv -> PCM
// Numerics: every product as in k_gemm_wres / k_resblock (activations hi + lo, bf16 weights, f32 accumulation, the same k order within
// each product), so the samples are the two-launch pipeline's, bit for bit.
#include <cstdlib>

#include "kernels.h"
#include "device_util.h"

namespace ptts {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2u(float a, float b, unsigned& hi, unsigned& lo) {
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);
    f32x2 r = f - __builtin_convertvector(h, f32x2);
    bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

union FragU {
    bf16x8 v;
    uint4 q;
    u32x4 w;
};

// f32 u tile [128 rows][64 channels]: 16-byte chunk k of row i sits at chunk k ^ usw(i).  Stage 0 writes rows 4 r + phase (r = lane & 15),
// stage C reads rows 16 wave + r: with this XOR both hit 16 different chunks per lane group.
__device__ __forceinline__ int usw(int i) { return (i ^ (i >> 2)) & 15; }
// the elu(u) / sum planes: k_resblock swizzles chunk k of plane row rho to k ^ rho, which suits lanes that hold CONSECUTIVE rows (its stages, and
// stages B .. D here).  Stage 0's lanes hold rows four apart (one phase of 16 consecutive input rows): with rho ^ (rho >> 2) instead, eight such
// rows hit eight different chunks as well (k ^ rho alone: two -- an 8-way bank conflict on every plane write of the stage).
__device__ __forceinline__ int esw(int rho) { return rho ^ (rho >> 2); }

}  // namespace

__global__ __launch_bounds__(512) void k_resblock_up(ResArgs a) {
    constexpr int C = 64, H = 32, CI = 128, NW = 8, NTH = NW * 64;
    constexpr int TR = NW * 16, HALO = 4, TOUT = TR - HALO;          // 128 output rows per tile, 124 of them new
    constexpr int TI = TR / 4, XR = TI + 1;                          // 32 input rows + the one before them
    constexpr int ROWB = C * 2;                                      // bytes per row of the elu(u) / sum planes (XOR-swizzled chunks: esw)
    // the input planes and the hidden planes are PADDED instead (a row pitch of 32 resp. 16 bytes more than the row, chunks in place): by the
    // lane groups a ds_read_b128 is served in, 16 consecutive rows then cover all banks whatever the first row (the XOR form was 2-way on
    // every read that starts at an odd row: the x[t] half of the window) -- tools/probes/lds_conflicts.py is the model these came from
    constexpr int HROWB = H * 2 + 16, XROWB = CI * 2 + 32;
    constexpr int CM = C / 8 - 1;                                    // chunk-swizzle mask
    constexpr int PLANE = (TR + 2) * ROWB, HPLANE = TR * HROWB, XPLANE = XR * XROWB;
    constexpr int F1 = (H / 16) * (3 * C / 32), F2 = (C / 16) * (H / 32), FF = 2 * (3 * C / 32);   // weight fragments (1 KiB each)
    constexpr int UT_BYTES = TR * C * 4;
    constexpr int OFF_EU = 2 * XPLANE, OFF_H = OFF_EU + 2 * PLANE, OFF_U = OFF_H + 2 * HPLANE, OFF_W = OFF_U + UT_BYTES;
    constexpr int OFF_B = OFF_W + (F1 + F2 + FF) * 1024;
    __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_B + (H + C + 4 + C) * 4];
    unsigned char* x_hi = smem;
    unsigned char* x_lo = smem + XPLANE;
    unsigned char* eu_hi = smem + OFF_EU;
    unsigned char* eu_lo = eu_hi + PLANE;
    unsigned char* h_hi = smem + OFF_H;
    unsigned char* h_lo = h_hi + HPLANE;
    unsigned char* ut = smem + OFF_U;
    const uint4* wl1 = reinterpret_cast<const uint4*>(smem + OFF_W);
    const uint4* wl2 = wl1 + F1 * 64;
    const uint4* wlf = wl2 + F2 * 64;
    float* bl1 = reinterpret_cast<float*>(smem + OFF_B);              // b1 [H], b2 [C], bf [1 (+3)], the transposed conv's bias [C]
    float* bl2 = bl1 + H;
    float* blf = bl2 + C;
    float* blu = blf + 4;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int tiles = (a.t1 - a.t0 + TOUT - 1) / TOUT;
    const int total = a.B * tiles;
    constexpr int XV = (XR * (CI / 4) + NTH - 1) / NTH;              // float4 of the tile's input rows per thread (3; the last partly used)
    float4 x[XV];
    auto request_rows = [&](int t) {                                   // input rows of tile t (clamped addresses; masked when consumed)
        const int bi_ = t / tiles, row0_ = a.t0 + (t % tiles) * TOUT - HALO;
        const int tin_ = (row0_ >> 2) - 1;                             // row0 is a multiple of 4 (host): input row of plane row 0
        const float* xb_ = a.xin + (int64_t)bi_ * a.x_bs + (int64_t)a.x_pad * CI;
#pragma unroll
        for (int j = 0; j < XV; j++) {
            const int e = min(tid + j * NTH, XR * (CI / 4) - 1), r = e / (CI / 4), c4 = e % (CI / 4);
            const int ti = min(max(tin_ + r, -a.x_pad), a.x_L - 1);
            x[j] = *reinterpret_cast<const float4*>(xb_ + (int64_t)ti * CI + c4 * 4);
        }
    };
    int tile = blockIdx.x;
    request_rows(tile);
    // this wave's share of the transposed convolution's weights: column tiles 2 wave, 2 wave + 1, all eight k steps, for good
    FragU wu[2][8];
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int s = 0; s < 8; s++) wu[ct][s].w = reinterpret_cast<const u32x4*>(a.wup)[((2 * wave + ct) * 8 + s) * 64 + lane];
    {
        uint4* dst = reinterpret_cast<uint4*>(smem + OFF_W);
        for (int i = tid; i < F1 * 64; i += NTH) dst[i] = reinterpret_cast<const uint4*>(a.w1)[i];
        for (int i = tid; i < F2 * 64; i += NTH) dst[F1 * 64 + i] = reinterpret_cast<const uint4*>(a.w2)[i];
        for (int i = tid; i < (FF / 2) * 64; i += NTH) {
            dst[(F1 + F2) * 64 + i] = reinterpret_cast<const uint4*>(a.wf_hi)[i];
            dst[(F1 + F2 + FF / 2) * 64 + i] = reinterpret_cast<const uint4*>(a.wf_lo)[i];
        }
        if (tid < H) bl1[tid] = a.b1 ? a.b1[tid] : 0.0f;
        if (tid < C) bl2[tid] = a.b2 ? a.b2[tid] : 0.0f;
        if (tid == 0) blf[0] = a.bf ? a.bf[0] : 0.0f;
        if (tid < C) blu[tid] = a.bup ? a.bup[tid] : 0.0f;            // (the same for every phase)
    }
    for (;;) {
        const int bi = tile / tiles, tb = a.t0 + (tile % tiles) * TOUT;   // first new output row of the tile
        const int row0 = tb - HALO;                                         // global output row of tile row 0
        const int tin = (row0 >> 2) - 1;                                    // input row of plane row 0
        PcmRow pr{nullptr, 0, 0};
        if (a.pcm_rows) pr = a.pcm_rows[bi];

        // ---- X: input rows -> hi / lo planes (rows past the utterance's end are zeros; rows before it are its zero history in memory) ----
#pragma unroll
        for (int j = 0; j < XV; j++) {
            const int e = tid + j * NTH;
            if (e < XR * (CI / 4)) {
                const int r = e / (CI / 4), c = (e % (CI / 4)) * 4;
                float4 v = x[j];
                if (tin + r >= a.x_L) v = make_float4(0.f, 0.f, 0.f, 0.f);
                unsigned h01, l01, h23, l23;
                split2u(v.x, v.y, h01, l01);
                split2u(v.z, v.w, h23, l23);
                const int off = r * XROWB + c * 2;
                *reinterpret_cast<uint2*>(x_hi + off) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(x_lo + off) = make_uint2(l01, l23);
            }
        }
        if (tid < 2 * ROWB / 16) {                    // the two rows in front of the tile only feed halo rows: zeros
            reinterpret_cast<uint4*>(eu_hi)[tid] = make_uint4(0, 0, 0, 0);
            reinterpret_cast<uint4*>(eu_lo)[tid] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        request_rows(min(tile + (int)gridDim.x, total - 1));   // past the block's last tile: a re-read that is never consumed

        // ---- 0: u = convtr(x): window of input row t = plane rows t, t + 1 (x[t-1] | x[t]); this wave's 32 columns, all 32 rows ----
        {
            f32x4 acc[2][2];
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
#pragma unroll
                for (int rt = 0; rt < 2; rt++) acc[ct][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const int jr = s >> 2, c0 = (s & 3) * 32;
#pragma unroll
                for (int rt = 0; rt < 2; rt++) {
                    const int rho = rt * 16 + r16 + jr;
                    const int off = rho * XROWB + ((c0 >> 3) + g) * 16;
                    FragU xh, xl;
                    xh.q = *reinterpret_cast<const uint4*>(x_hi + off);
                    xl.q = *reinterpret_cast<const uint4*>(x_lo + off);
#pragma unroll
                    for (int ct = 0; ct < 2; ct++) {
                        acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wu[ct][s].v, xh.v, acc[ct][rt], 0, 0, 0);
                        PTTS_LO_MFMA(acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wu[ct][s].v, xl.v, acc[ct][rt], 0, 0, 0));
                    }
                }
            }
            // lane: input row rt*16 + r16; the weight columns are grouped (model.cpp) so that columns 4 g .. 4 g + 3 of tile (wave, ct) are phase g of
            // channels 8 wave + 4 ct .. + 3: the lane's four values are four channels of output row 4 (input row) + g, and a wave's 64 lanes
            // hold 64 consecutive output rows
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const int i = 4 * (rt * 16 + r16) + g;         // tile row of the output
                const int gr = row0 + i;
                const bool ok = gr >= 0 && gr < a.L;
                const int rho = i + 2;
#pragma unroll
                for (int ct = 0; ct < 2; ct++) {
                    const int ch = 8 * wave + 4 * ct;
                    const float4 b = *reinterpret_cast<const float4*>(blu + ch);
                    float4 v = make_float4(acc[ct][rt][0] + b.x, acc[ct][rt][1] + b.y, acc[ct][rt][2] + b.z, acc[ct][rt][3] + b.w);
                    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);   // rows before the utterance (zero history) or past its end
                    *reinterpret_cast<float4*>(ut + i * (C * 4) + (((ch >> 2) ^ usw(i)) << 4)) = v;
                    unsigned h01, l01, h23, l23;
                    split2u(elu_fast(v.x), elu_fast(v.y), h01, l01);
                    split2u(elu_fast(v.z), elu_fast(v.w), h23, l23);
                    const int off = rho * ROWB + ((((ch >> 3) ^ esw(rho)) & CM) << 4) + ((ch & 4) << 1);
                    *reinterpret_cast<uint2*>(eu_hi + off) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2*>(eu_lo + off) = make_uint2(l01, l23);
                }
            }
        }
        __syncthreads();

        const int i_lane = wave * 16 + r16;               // tile row this lane owns in the MFMA operands / results of stages B .. D
        const int gr_lane = row0 + i_lane;
        const bool in_seq = gr_lane >= 0 && gr_lane < a.L;
        // ---- B: hidden = elu(conv_k3(eu) + b1) ----
        {
            constexpr int NT = H / 16, KS = 3 * C / 32;
            f32x4 acc[NT];
#pragma unroll
            for (int n = 0; n < NT; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const uint4* w1 = wl1 + lane;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const int tap = (s * 32) / C, c0 = (s * 32) % C;
                const int rho = i_lane + tap;             // window row i-2+tap, stored at rho = that + 2
                const int off = rho * ROWB + ((((c0 >> 3) + g) ^ esw(rho)) & CM) * 16;
                FragU xh, xl;
                xh.q = *reinterpret_cast<const uint4*>(eu_hi + off);
                xl.q = *reinterpret_cast<const uint4*>(eu_lo + off);
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    FragU wh;
                    wh.q = w1[(n * KS + s) * 64];
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xh.v, acc[n], 0, 0, 0);
                    PTTS_LO_MFMA(acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xl.v, acc[n], 0, 0, 0));
                }
            }
#pragma unroll
            for (int n = 0; n < NT; n++) {                // lane: row i_lane, hidden channels n*16 + 4g .. +3
                const int ch = n * 16 + 4 * g;
                const float4 b = *reinterpret_cast<const float4*>(bl1 + ch);
                unsigned h01, l01, h23, l23;
                split2u(elu_fast(acc[n][0] + b.x), elu_fast(acc[n][1] + b.y), h01, l01);
                split2u(elu_fast(acc[n][2] + b.z), elu_fast(acc[n][3] + b.w), h23, l23);
                const int off = i_lane * HROWB + ch * 2;
                *reinterpret_cast<uint2*>(h_hi + off) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(h_lo + off) = make_uint2(l01, l23);
            }
        }
        __syncthreads();

        // ---- C: sum = elu(u + conv_k1(hidden) + b2) -> planes (elu(u) is dead since stage B's barrier) ----
        {
            constexpr int NT = C / 16, KS = H / 32;
            f32x4 acc[NT];
#pragma unroll
            for (int n = 0; n < NT; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const uint4* w2 = wl2 + lane;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const int off = i_lane * HROWB + (s * 4 + g) * 16;
                FragU xh, xl;
                xh.q = *reinterpret_cast<const uint4*>(h_hi + off);
                xl.q = *reinterpret_cast<const uint4*>(h_lo + off);
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    FragU wh;
                    wh.q = w2[(n * KS + s) * 64];
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xh.v, acc[n], 0, 0, 0);
                    PTTS_LO_MFMA(acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xl.v, acc[n], 0, 0, 0));
                }
            }
#pragma unroll
            for (int n = 0; n < NT; n++) {
                const int ch = n * 16 + 4 * g;
                const float4 b = *reinterpret_cast<const float4*>(bl2 + ch);
                const float4 ur = *reinterpret_cast<const float4*>(ut + i_lane * (C * 4) + (((ch >> 2) ^ usw(i_lane)) << 4));
                float4 v;
                v.x = elu_fast(ur.x + (acc[n][0] + b.x)); v.y = elu_fast(ur.y + (acc[n][1] + b.y));
                v.z = elu_fast(ur.z + (acc[n][2] + b.z)); v.w = elu_fast(ur.w + (acc[n][3] + b.w));
                if (!in_seq) v = make_float4(0.f, 0.f, 0.f, 0.f);   // rows before the utterance are the final conv's zero padding
                const int rho = i_lane + 2;
                unsigned h01, l01, h23, l23;
                split2u(v.x, v.y, h01, l01);
                split2u(v.z, v.w, h23, l23);
                const int off = rho * ROWB + ((((ch >> 3) ^ esw(rho)) & CM) << 4) + ((ch & 4) << 1);
                *reinterpret_cast<uint2*>(eu_hi + off) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(eu_lo + off) = make_uint2(l01, l23);
            }
        }
        // ---- D: pcm[row] = bf + sum_{tap, c} sum[row-2+tap][c] * wf[tap*C + c] ----
        __syncthreads();
        {
            constexpr int KS = 3 * C / 32;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const uint4* wfh = wlf + lane;
            const uint4* wfl = wlf + (FF / 2) * 64 + lane;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const int tap = (s * 32) / C, c0 = (s * 32) % C;
                const int rho = i_lane + tap;
                const int off = rho * ROWB + ((((c0 >> 3) + g) ^ esw(rho)) & CM) * 16;
                FragU xh, xl, wh, wl;
                xh.q = *reinterpret_cast<const uint4*>(eu_hi + off);
                xl.q = *reinterpret_cast<const uint4*>(eu_lo + off);
                wh.q = wfh[s * 64];
                wl.q = wfl[s * 64];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xh.v, acc, 0, 0, 0);
                PTTS_LO_MFMA(acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, xl.v, acc, 0, 0, 0));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl.v, xh.v, acc, 0, 0, 0);
            }
            const int gr = row0 + i_lane;                 // result column 0 sits in register 0 of lane group 0
            const float smp = acc[0] + blf[0];
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
        tile += gridDim.x;
        if (tile >= total) break;
        __syncthreads();   // the planes, the u tile and the staging rows are rewritten by the next tile
    }
}

// the fused form takes what k_resblock<64, 32, 8, FINAL, bf16, persistent> takes, plus the transposed convolution in front of it
bool resblock_up_supported(const ResArgs& a) {
    if (!(a.fuse_up && a.xin && a.wup && a.final_conv && a.w_bf16 && a.C == 64 && a.H == 32 && a.CI == 128 && a.up_stride == 4)) return false;
    if (!resblock_supported(a) || a.x_pad < 1 || a.t0 % 4 != 0 || a.x_L * 4 != a.L || !aligned16(a.xin) || a.x_bs % 4 != 0) return false;
    static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const int tiles = (a.t1 - a.t0 + 124 - 1) / 124;
    return a.B * tiles >= cus * 8;                       // enough tiles per block to amortise its weight copies (156 KB)
}

void launch_resblock_up(const ResArgs& a, hipStream_t stream) {
    note_launch("k_resblock_up+final");
    static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    hipLaunchKernelGGL(k_resblock_up, dim3((unsigned)cus), dim3(512), 0, stream, a);   // 128 KB of LDS: one block per CU
}

}  // namespace ptts
