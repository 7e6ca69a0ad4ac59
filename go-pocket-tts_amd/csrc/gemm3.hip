// gemm3.hip -- the many-row GEMM of the Mimi decoder and of prefill, second generation (every Linear / Conv1d /
// ConvTranspose1d of mimi.go:719-789 and flow_transformer.go:749-771 at M = thousands .. millions of rows).
#include "kernels.h"
#include "device_util.h"

namespace ptts {

// C[M,N] = epi( aop(A)[M,K] * W[N,K]^T ), same numerics as k_gemm2 (activations split into bf16 hi + lo, bf16 weights
// multiply both, f32 weights are split too and take three products, f32 accumulation), different data movement:
//
//  * the activations never touch LDS.  The A operand of v_mfma_f32_16x16x32_bf16 wants, in lane l, eight consecutive k of
//    row l&15 starting at 8*(l>>4) -- i.e. 32 contiguous bytes of an f32 row, and the four lane groups of a row together
//    read 128 contiguous bytes.  Each wave therefore loads its own 32 rows straight from HBM/L2 with 16-byte loads that
//    use every byte of every sector, splits them in registers (v_cvt_pk_bf16_f32) and feeds the matrix core.  k_gemm2
//    staged hi and lo through LDS, which cost 3x the LDS traffic of the weights and bounded the kernel at the LDS port;
//  * only the weight tile (BN columns x 64 k, shared by the 8 waves of the block) goes through LDS, double-buffered, one
//    barrier per 64 k, laid out [k-half][column][32 k] so that a fragment read is one contiguous 1-KiB ds_read_b128;
//  * the product is computed transposed (weights as the MFMA's row operand): a lane ends up with FOUR CONSECUTIVE output
//    columns of one row, so bias / residual / store are 16-byte accesses.
// Block = 8 waves x 32 rows = 256 rows x BN columns (BN = 128 / 64 / 32 by N); per 32 k a wave issues 4 global loads,
// BN/16 LDS reads and BN/4 (bf16 weights) MFMAs.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2h(float a, float b, unsigned& hi, unsigned& lo) {
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);
    f32x2 r = f - __builtin_convertvector(h, f32x2);
    bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

union Frag3 {
    bf16x8 v;
    uint4 q;
};

thread_local int g_gemm3_cfg = 0;   // debug knob (ptts_debug_gemm): 0 = default shape

template <int BN, bool WBF16, int NW, int CH>
__global__ __launch_bounds__(NW * 64) void k_gemm3(GemmArgs a) {
    constexpr int G3_BM = NW * 32, G3_CH = CH, NTH = NW * 64, SPC = CH / 32;   // rows per block, k per weight chunk, threads, 32-k steps per chunk
    constexpr int NT = BN / 16;
    constexpr int HALF = BN * 64;                         // bytes of one [column][32 k] bf16 sub-chunk
    constexpr int PLANE = SPC * HALF;                       // hi (or only) plane of a stage
    constexpr int STAGE = WBF16 ? PLANE : 2 * PLANE;      // f32 weights: hi plane + lo plane
    constexpr int PPR = WBF16 ? CH / 8 : CH / 4;          // 16-byte global pieces per weight row and chunk
    constexpr int PIECES = BN * PPR;
    constexpr int PPT = (PIECES + NTH - 1) / NTH;
    __shared__ __attribute__((aligned(16))) char Ws[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    // XCD-aware order (speed only): all column tiles of a 256-row panel run on the XCD that already holds the panel in L2
    const int ncol = (a.N + BN - 1) / BN, npan = (a.M + G3_BM - 1) / G3_BM;
    const int bid = blockIdx.x, xcd = bid & 7, jb = bid >> 3;
    const int pan = (jb / ncol) * 8 + xcd;
    if (pan >= npan) return;
    const int m0 = pan * G3_BM + wave * 32, n0 = (jb % ncol) * BN;
    // split-K (a.kslice > 0): blockIdx.y owns k in [kz, kz + KL) and writes its raw sums to plane blockIdx.y of C
    const int kz = blockIdx.y * a.kslice;
    const int KL = a.kslice ? min(a.kslice, a.K - kz) : a.K;

    const float* aptr[2];
#pragma unroll
    for (int t = 0; t < 2; t++) aptr[t] = a.A + row_off(a.amap, min(m0 + t * 16 + r16, a.M - 1)) + kz + g * 8;

    uint4 wreg[PPT];
    auto w_load = [&](int c) {
        const int k0 = c * G3_CH;
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            const int pc = tid + NTH * p;
            const int n = pc / PPR;
            const int k = k0 + (pc % PPR) * (WBF16 ? 8 : 4);
            uint4 u = make_uint4(0, 0, 0, 0);
            if (pc < PIECES && n0 + n < a.N && k < KL)
                u = *reinterpret_cast<const uint4*>((const char*)a.W + ((int64_t)(n0 + n) * a.ldw + kz + k) * (WBF16 ? 2 : 4));
            wreg[p] = u;
        }
    };
    auto w_store = [&](int stage) {
        char* base = Ws + stage * STAGE;
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            const int pc = tid + NTH * p;
            if (pc >= PIECES) continue;
            const int n = pc / PPR, kk = pc % PPR;
            if constexpr (WBF16) {   // 8 k at kk*8: sub-chunk kk>>2, 16-byte slot kk&3
                *reinterpret_cast<uint4*>(base + (kk >> 2) * HALF + n * 64 + (kk & 3) * 16) = wreg[p];
            } else {                 // 4 k at kk*4: sub-chunk kk>>3, 8-byte slot kk&7
                unsigned h01, l01, h23, l23;
                split2h(__uint_as_float(wreg[p].x), __uint_as_float(wreg[p].y), h01, l01);
                split2h(__uint_as_float(wreg[p].z), __uint_as_float(wreg[p].w), h23, l23);
                const int off = (kk >> 3) * HALF + n * 64 + (kk & 7) * 8;
                *reinterpret_cast<uint2*>(base + off) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(base + PLANE + off) = make_uint2(l01, l23);
            }
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int n = 0; n < NT; n++) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = KL >> 5, nchunks = (KL + G3_CH - 1) / G3_CH;
    // activations: a ring of SPC 32-k steps per wave.  The slot of step i is refilled with step i + SPC as soon as step i
    // has been split into fragments, so every wave keeps SPC steps (CH k of its 32 rows) in flight at all times.
    float4 av[SPC][2][2];
    auto a_load = [&](int i, float4 (&dst)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            dst[t][0] = *reinterpret_cast<const float4*>(aptr[t] + i * 32);
            dst[t][1] = *reinterpret_cast<const float4*>(aptr[t] + i * 32 + 4);
        }
    };
    w_load(0);
#pragma unroll
    for (int s = 0; s < SPC; s++)
        if (s < nsteps) a_load(s, av[s]);
    w_store(0);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        if (c + 1 < nchunks) w_load(c + 1);
#pragma unroll
        for (int s = 0; s < SPC; s++) {
            const int i = c * SPC + s;
            if (i < nsteps) {
                Frag3 ah[2], al[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    float4 x0 = av[s][t][0], x1 = av[s][t][1];
                    if (a.aop == AOP_ELU) {
                        x0.x = elu_fast(x0.x); x0.y = elu_fast(x0.y); x0.z = elu_fast(x0.z); x0.w = elu_fast(x0.w);
                        x1.x = elu_fast(x1.x); x1.y = elu_fast(x1.y); x1.z = elu_fast(x1.z); x1.w = elu_fast(x1.w);
                    }
                    split2h(x0.x, x0.y, ah[t].q.x, al[t].q.x);
                    split2h(x0.z, x0.w, ah[t].q.y, al[t].q.y);
                    split2h(x1.x, x1.y, ah[t].q.z, al[t].q.z);
                    split2h(x1.z, x1.w, ah[t].q.w, al[t].q.w);
                }
                if (i + SPC < nsteps) a_load(i + SPC, av[s]);
                const char* wb = Ws + (c & 1) * STAGE + s * HALF + r16 * 64 + g * 16;
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    Frag3 wh;
                    wh.q = *reinterpret_cast<const uint4*>(wb + n * 1024);
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, ah[t].v, acc[t][n], 0, 0, 0);
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh.v, al[t].v, acc[t][n], 0, 0, 0);
                    }
                    if constexpr (!WBF16) {
                        Frag3 wl;
                        wl.q = *reinterpret_cast<const uint4*>(wb + PLANE + n * 1024);
#pragma unroll
                        for (int t = 0; t < 2; t++) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl.v, ah[t].v, acc[t][n], 0, 0, 0);
                    }
                }
            }
        }
        if (c + 1 < nchunks) w_store((c + 1) & 1);
        __syncthreads();
    }

    // epilogue: lane holds C[row = tile row r16][columns nt*16 + 4*g .. +3]
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int m = m0 + t * 16 + r16;
        if (m >= a.M) continue;
        const int64_t ro = row_off(a.cmap, m) + blockIdx.y * a.zstride;
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const int col = n0 + n * 16 + 4 * g;
            if (col >= a.N) continue;   // N % 4 == 0
            float4 v = make_float4(acc[t][n][0], acc[t][n][1], acc[t][n][2], acc[t][n][3]);
            if (a.bias) {
                const float4 b = *reinterpret_cast<const float4*>(a.bias + col);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            if (a.rope_cos && col < a.rope_cols) {   // the lane's four columns are two (even, odd) pairs of one head
                const int pos = a.rope_row_pos ? a.rope_row_pos[m] : a.rope_pos0 + (a.rope_rows_per_seg ? m % a.rope_rows_per_seg : m);
                const int half = a.rope_hd >> 1, j = (col % a.rope_hd) >> 1;
                const float2 cs = *reinterpret_cast<const float2*>(a.rope_cos + (int64_t)pos * half + j);
                const float2 sn = *reinterpret_cast<const float2*>(a.rope_sin + (int64_t)pos * half + j);
                const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
                {
#pragma clang fp contract(off)
                    // rounded product by product (the reference's x*c - y*s, rope.go:81-105), the same bits as k_gemm5's epilogue
                    v.x = x0 * cs.x - x1 * sn.x; v.y = x0 * sn.x + x1 * cs.x;
                    v.z = x2 * cs.y - x3 * sn.y; v.w = x2 * sn.y + x3 * cs.y;
                }
            }
            const int64_t co = ro + col;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.epi >= EPI_RESADD) r = *reinterpret_cast<const float4*>(a.R + co);
            {
#pragma clang fp contract(off)
            // one rounding per operation, like k_gemm5's epilogue and like the reference (r + s*v, r + alpha*v on amd64): with contraction the
            // same row could get different last bits in different kernels
            switch (a.epi) {
                case EPI_NONE: break;
                case EPI_GELU: v.x = gelu1(v.x); v.y = gelu1(v.y); v.z = gelu1(v.z); v.w = gelu1(v.w); break;
                case EPI_SILU: {
                    float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.addvec) ad = *reinterpret_cast<const float4*>(a.addvec + col);
                    v.x = silu1(ad.x + v.x); v.y = silu1(ad.y + v.y); v.z = silu1(ad.z + v.z); v.w = silu1(ad.w + v.w);
                    break;
                }
                case EPI_ELU: v.x = elu_fast(v.x); v.y = elu_fast(v.y); v.z = elu_fast(v.z); v.w = elu_fast(v.w); break;
                case EPI_RESADD: v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w; break;
                case EPI_SCALE_RESADD: {
                    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (a.scale) sc = *reinterpret_cast<const float4*>(a.scale + col);
                    v.x = r.x + sc.x * v.x; v.y = r.y + sc.y * v.y; v.z = r.z + sc.z * v.z; v.w = r.w + sc.w * v.w;
                    break;
                }
                case EPI_GATE_RESADD: {
                    const float4 gt = *reinterpret_cast<const float4*>(a.gate + (int64_t)m * a.ldg + col);
                    v.x = r.x + gt.x * v.x; v.y = r.y + gt.y * v.y; v.z = r.z + gt.z * v.z; v.w = r.w + gt.w * v.w;
                    break;
                }
                case EPI_RESADD_ELU: v.x = elu_fast(r.x + v.x); v.y = elu_fast(r.y + v.y); v.z = elu_fast(r.z + v.z); v.w = elu_fast(r.w + v.w); break;
                case EPI_AXPY: v.x = r.x + a.alpha * v.x; v.y = r.y + a.alpha * v.y; v.z = r.z + a.alpha * v.z; v.w = r.w + a.alpha * v.w; break;
            }
            }
            *reinterpret_cast<float4*>(a.C + co) = v;
        }
    }
}

bool gemm3_supported(const GemmArgs& a) {
    const int kalign = a.w_bf16 ? 8 : 4;
    const bool res = a.epi >= EPI_RESADD;
    return a.M >= 512 && a.K % 32 == 0 && a.N % 4 == 0 && aligned16(a.A) && a.amap.ld % 4 == 0 && a.amap.batch_stride % 4 == 0 &&
           a.ldw % kalign == 0 && aligned16(a.W) && aligned16(a.C) && a.cmap.ld % 4 == 0 && a.cmap.batch_stride % 4 == 0 &&
           (!a.bias || aligned16(a.bias)) && (!a.addvec || aligned16(a.addvec)) && (!a.scale || aligned16(a.scale)) &&
           (!res || aligned16(a.R)) && (a.epi != EPI_GATE_RESADD || (aligned16(a.gate) && a.ldg % 4 == 0)) &&
           (!a.rope_cos || (a.rope_hd % 4 == 0 && a.rope_cols % 4 == 0 && a.epi == EPI_NONE)) &&
           (!a.kslice || (a.kslice % 128 == 0 && a.epi == EPI_NONE && !a.bias && !a.rope_cos && a.zstride % 4 == 0));
}

template <int BN, int NW, int CH>
static void launch3_cfg(const GemmArgs& a, hipStream_t stream) {
    const int ncol = (a.N + BN - 1) / BN, npan = (a.M + NW * 32 - 1) / (NW * 32);
    dim3 grid((unsigned)(((npan + 7) / 8) * 8 * ncol), (unsigned)(a.kslice ? (a.K + a.kslice - 1) / a.kslice : 1));
    if (a.w_bf16) hipLaunchKernelGGL((k_gemm3<BN, true, NW, CH>), grid, dim3(NW * 64), 0, stream, a);
    else hipLaunchKernelGGL((k_gemm3<BN, false, NW, CH>), grid, dim3(NW * 64), 0, stream, a);
}

template <int BN>
static void launch3_bn(const GemmArgs& a, hipStream_t stream) {
    switch (g_gemm3_cfg) {   // 0: shape picked from M; the rest are for tools/microbench_gemm.py
        case 1: launch3_cfg<BN, 4, 64>(a, stream); break;
        case 2: launch3_cfg<BN, 4, 128>(a, stream); break;
        case 3: launch3_cfg<BN, 8, 64>(a, stream); break;
        case 6: launch3_cfg<BN, 8, 128>(a, stream); break;
        default:
            if (a.M < 16384) launch3_cfg<BN, 4, 64>(a, stream);   // few panels: 128-row blocks, up to 3 per CU
            else launch3_cfg<BN, 8, 128>(a, stream);
    }
}

// Measured on MI355X (tools/microbench_gemm.py, B=64 Mimi shapes, bf16 weights): 256 columns per wave halve the activation
// loads per MFMA and win from N = 512 up (K = 512: +7..17 %, K >= 1024: +21..29 %); at N = 256 the two are level.
void launch_gemm3(const GemmArgs& a, hipStream_t stream) {
    note_launch(a.rope_cos ? "k_gemm3+rope" : a.kslice ? "k_gemm3+splitk" : "k_gemm3");
    const bool wide = g_gemm3_cfg == 4 || (g_gemm3_cfg == 0 && a.N >= 512 && a.M >= 16384);
    // few row panels (the prompt prefill: ~1600 rows): 128-column blocks would cover under 80 % of the CUs -- halve the block's columns
    const bool narrow = g_gemm3_cfg == 0 && a.M < 16384 && ((a.M + 127) / 128) * ((a.N + 127) / 128) * (a.kslice ? (a.K + a.kslice - 1) / a.kslice : 1) < 200;
    if (wide && a.N >= 256) launch3_cfg<256, 8, 64>(a, stream);
    else if (a.N > 64 && !narrow) launch3_bn<128>(a, stream);
    else if (a.N > 32) launch3_bn<64>(a, stream);
    else launch3_bn<32>(a, stream);
}

}  // namespace ptts
