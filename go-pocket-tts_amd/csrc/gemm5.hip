// gemm5.hip -- the many-row GEMM, third generation (bf16 weights, M >= 1024 rows: the decoder transformer's linears, the first
// SEANet convolution and transposed convolution, the wide residual block -- mimi.go:719-789, conv1d.go:20-83,
// convtranspose1d.go:73-148 -- and the prompt prefill's projections, flow_transformer.go:749-771, in 128 x 128 tiles).
//
// Data movement and numerics are k_gemm3's (gemm3.hip): every wave loads its own 32 rows of f32 activations in full 128-byte
// lines straight into registers, splits them into bf16 hi + lo and multiplies both by the bf16 weights
// (v_mfma_f32_16x16x32_bf16, f32 accumulation, the same k order -> the same bits); only the weight chunk shared by the block's
// waves goes through LDS.  What changed is everything around the products, found by reading k_gemm3's ISA and its stamps:
//   * no load sits behind a runtime condition.  k_gemm3 predicated its weight loads (column / k bounds) and chose the epilogue
//     form per value inside the unrolled store loop: hipcc branches around such loads and waits vmcnt(0) at the joins, which
//     drained the activation ring once per weight chunk and made the epilogue 32 dependent round trips (38 000 cycles per tile
//     beside a 64 000-cycle K loop at K = 512).  Here addresses are clamped instead, bounds are applied to the STORES, the
//     epilogue form is decided once, and its operands (residual, bias, RoPE table rows) are requested eight at a time;
//   * the weight image in LDS is XOR-swizzled for the lane groups gfx950 serves a ds_read_b128 in (every fragment read was a
//     2-way bank conflict in k_gemm3's layout, and the fragment reads are half of the LDS port's time at full MFMA rate);
//   * the prologue ELU of a consumer (the residual blocks' first convolution) is a template parameter, not a branch in the loop.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "device_util.h"

namespace ptts {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // (a uint4 copied out of global memory is a struct memcpy the optimiser leaves in scratch)

union Frag5 {
    bf16x8 v;
    uint4 q;
};

__device__ __forceinline__ void split5(float a, float b, unsigned& hi, unsigned& lo) {
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);
    f32x2 r = f - __builtin_convertvector(h, f32x2);
    bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

// A weight column's 32 k (64 bytes = four 16-byte chunks) are stored with chunk c at c ^ fw(column).  A ds_read_b128 is served
// in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32: each group holds every fragment column once,
// columns 4..11 with the neighbouring k group; with this XOR its 16 lanes hit 16 different 16-byte bank slots.
__device__ __forceinline__ int fw5(int col) { return ((col >> 3) & 1) * 3; }

template <int V> using ic = std::integral_constant<int, V>;

}  // namespace

thread_local int g_gemm5_cfg = 0;   // debug knob (ptts_debug_gemm): 0 = default shape

// ABL (measurement builds only): 1 no activation loads in the loop, 2 one fragment read per step, 4 no split, 8 no stores, 16 no MFMA
template <int BN, int NW, bool ELU_A, int ABL>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm5(GemmArgs a) {
    constexpr int BM = NW * 32, CH = 64, SPC = CH / 32, NTH = NW * 64, NT = BN / 16;
    // bytes: one [column][32 k] sub-chunk (+ 64 of padding: the two sub-chunks of a column are written by one ds_write_b128 service group of 8 lanes --
    // four lanes each -- and at a distance of BN * 64 they sat on the SAME banks: a 2-way conflict on every weight store, which is what the
    // SQ_LDS_BANK_CONFLICT counter showed for this kernel in round 3 while the fragment READS were conflict-free), one stage
    constexpr int HALF = BN * 64 + 64, STAGE = SPC * HALF;
    constexpr int PPR = CH / 8, PIECES = BN * PPR, PPT = PIECES / NTH;
    static_assert(PIECES % NTH == 0, "whole 16-byte pieces per thread");
    __shared__ __attribute__((aligned(16))) char Ws[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    // XCD-aware order (speed only): all column tiles of a row panel run on the XCD that already holds the panel in L2
    const int ncol = (a.N + BN - 1) / BN, npan = (a.M + BM - 1) / BM;
    const int bid = blockIdx.x, xcd = bid & 7, jb = bid >> 3;
    const int pan = (jb / ncol) * 8 + xcd;
    if (pan >= npan) return;
    const int m0 = pan * BM + wave * 32, n0 = (jb % ncol) * BN;

    // split-K (a.kslice > 0, the prompt prefill's linear2): blockIdx.y owns k in [kz, kz + KL) and writes its raw sums to plane blockIdx.y of C
    const int kz = blockIdx.y * a.kslice;
    const int KL = a.kslice ? min(a.kslice, a.K - kz) : a.K;
    const float* aptr[2];
#pragma unroll
    for (int t = 0; t < 2; t++) aptr[t] = a.A + row_off(a.amap, min(m0 + t * 16 + r16, a.M - 1)) + kz + g * 8;
    // weights: thread -> (column tid / 8 + 64 NW/8 p, k piece tid % 8); N % BN == 0 (host), so no column needs a bound
    constexpr int CPP = NTH / PPR;                                       // columns between two pieces of a thread
    const int wn = tid / PPR, wk = tid % PPR;                            // 8 k at wk*8: sub-chunk wk>>2, chunk wk&3
    const char* wsrc = (const char*)a.W + ((int64_t)(n0 + wn) * a.ldw + kz + wk * 8) * 2;
    const int64_t wstep = (int64_t)CPP * a.ldw * 2;
    const int wdst = (wk >> 2) * HALF + wn * 64 + (((wk & 3) ^ fw5(wn)) << 4);   // CPP % 16 == 0: the swizzle of column wn + CPP p is wn's
    static_assert(CPP % 16 == 0, "");
    u32x4 wreg[PPT];
    // k offset of the c-th 64-deep chunk.  A causal convolution as a product reads, for output row r, the window rows r-taps+1 .. r (K = taps x C,
    // tap-major in memory and in the weights).  Walked in that order a wave meets an input row again C / 64 chunks later -- once per tap -- and by
    // then the line has left L1 and mostly L2: the first convolution (7 taps) fetched its input 6x from HBM (FETCH_SIZE, profiles/r3_pmc_mimi.txt).
    // With GemmArgs::win_taps the chunks are walked channel-block-major instead -- (block 0: tap 0, 1, .. ), (block 1: ..) -- so the taps' reads of one
    // line follow each other.  Sums are per k chunk either way: the order of the chunks changes the last bits, not the products.
    const int wtaps = a.win_taps;
    auto koff = [&](int c) { return wtaps ? (c % wtaps) * a.win_c + (c / wtaps) * CH : c * CH; };
    auto w_load = [&](int c) {
        const int64_t ko = (int64_t)koff(c) * 2;
#pragma unroll
        for (int p = 0; p < PPT; p++) wreg[p] = *reinterpret_cast<const u32x4*>(wsrc + p * wstep + ko);
    };
    auto w_store = [&](int stage) {
#pragma unroll
        for (int p = 0; p < PPT; p++) *reinterpret_cast<u32x4*>(Ws + stage * STAGE + wdst + p * (CPP * 64)) = wreg[p];
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int n = 0; n < NT; n++) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = KL / CH, nsteps = nchunks * SPC;       // K (and a K slice) % CH == 0 (host)
    float4 av[SPC][2][2];
    auto a_load = [&](int i, float4 (&dst)[2][2]) {
        const int ko = koff(i / SPC) + (i % SPC) * 32;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            dst[t][0] = *reinterpret_cast<const float4*>(aptr[t] + ko);
            dst[t][1] = *reinterpret_cast<const float4*>(aptr[t] + ko + 4);
        }
    };
    w_load(0);
#pragma unroll
    for (int s = 0; s < SPC; s++) a_load(min(s, nsteps - 1), av[s]);
    w_store(0);
    __syncthreads();
    const int frag_off = r16 * 64 + ((g ^ fw5(r16)) << 4);
    for (int c = 0; c < nchunks; c++) {
        w_load(min(c + 1, nchunks - 1));                       // the last chunk is fetched (and staged) once more: no branch, nobody reads it
#pragma unroll
        for (int s = 0; s < SPC; s++) {
            const int i = c * SPC + s;
            Frag5 ah[2], al[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                float4 x0 = av[s][t][0], x1 = av[s][t][1];
                if constexpr (ELU_A) {
                    x0.x = elu_fast(x0.x); x0.y = elu_fast(x0.y); x0.z = elu_fast(x0.z); x0.w = elu_fast(x0.w);
                    x1.x = elu_fast(x1.x); x1.y = elu_fast(x1.y); x1.z = elu_fast(x1.z); x1.w = elu_fast(x1.w);
                }
                if constexpr (ABL & 4) {
                    ah[t].q = make_uint4(__float_as_uint(x0.x), __float_as_uint(x0.y), __float_as_uint(x0.z), __float_as_uint(x0.w));
                    al[t].q = make_uint4(__float_as_uint(x1.x), __float_as_uint(x1.y), __float_as_uint(x1.z), __float_as_uint(x1.w));
                } else {
                    split5(x0.x, x0.y, ah[t].q.x, al[t].q.x);
                    split5(x0.z, x0.w, ah[t].q.y, al[t].q.y);
                    split5(x1.x, x1.y, ah[t].q.z, al[t].q.z);
                    split5(x1.z, x1.w, ah[t].q.w, al[t].q.w);
                }
            }
            if constexpr (!(ABL & 1)) a_load(min(i + SPC, nsteps - 1), av[s]);
            __builtin_amdgcn_sched_barrier(0);                    // the loads stay in front of the products they run under
            const char* wb = Ws + (c & 1) * STAGE + s * HALF + frag_off;
            Frag5 wh[2];                                            // fragment n + 1 is requested before fragment n is multiplied
            wh[0].q = *reinterpret_cast<const uint4*>(wb);
            if constexpr (!(ABL & 18)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
            for (int n = 0; n < NT; n++) {
                if constexpr (!(ABL & 2)) {
                    if (n + 1 < NT) wh[(n + 1) & 1].q = *reinterpret_cast<const uint4*>(wb + (n + 1) * 1024);
                }
                const Frag5& w = (ABL & 2) ? wh[0] : wh[n & 1];
                if constexpr (!(ABL & 16)) {
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, ah[t].v, acc[t][n], 0, 0, 0);
                        PTTS_LO_MFMA(acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, al[t].v, acc[t][n], 0, 0, 0));
                    }
                } else {
                    acc[0][n][0] += __uint_as_float(w.q.x ^ ah[0].q.x ^ al[1].q.y);
                    acc[1][n][0] += __uint_as_float(w.q.y ^ ah[1].q.x ^ al[0].q.y);
                }
                if constexpr (!(ABL & 18)) {   // pin the order: the read of fragment n + 1, then the four products of fragment n
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        w_store((c + 1) & 1);
        __syncthreads();
    }

    // epilogue: lane holds C[row = tile row r16][columns n*16 + 4*g .. +3].  One form per launch, decided here; its operands
    // are requested eight column groups at a time, from clamped addresses, and only the stores are bounded.
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool has_bias = a.bias != nullptr;
    const float* bias_p = has_bias ? a.bias : reinterpret_cast<const float*>(a.W);   // any readable address: the value is discarded
    auto epilogue = [&](auto epi_c, auto rope_c) {
#pragma clang fp contract(off)
        constexpr int EPI = decltype(epi_c)::value;
        constexpr bool ROPE = decltype(rope_c)::value != 0;
        constexpr bool RES = EPI >= EPI_RESADD;
        constexpr int GRP = 4;
        const bool has_scale = EPI == EPI_SCALE_RESADD && a.scale != nullptr;
        const float* scale_p = has_scale ? a.scale : reinterpret_cast<const float*>(a.W);
        const int half = a.rope_hd >> 1;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int m = m0 + t * 16 + r16, mc = min(m, a.M - 1);
            const int64_t ro = row_off(a.cmap, mc) + blockIdx.y * a.zstride;
            int64_t tab = 0;
            if constexpr (ROPE) {
                const int pos = a.rope_row_pos ? a.rope_row_pos[mc] : a.rope_pos0 + (a.rope_rows_per_seg ? mc % a.rope_rows_per_seg : mc);
                tab = (int64_t)pos * half;
            }
            // RoPE table entries: column n*16 + 4g sits at pair index ((n & 3) * 8 + 2g) of its 64-wide head, so the four column
            // groups j = n & 3 of this lane repeat for every nb -- 2 x 4 loads per row instead of 2 x 16 (hd == 64: host)
            float2 cs[GRP], sn[GRP];
            if constexpr (ROPE) {
#pragma unroll
                for (int j = 0; j < GRP; j++) {
                    const int jj = ((j * 16 + 4 * g) & 63) >> 1;
                    cs[j] = *reinterpret_cast<const float2*>(a.rope_cos + tab + jj);
                    sn[j] = *reinterpret_cast<const float2*>(a.rope_sin + tab + jj);
                }
            }
#pragma unroll
            for (int nb = 0; nb < NT / GRP; nb++) {
                float4 rr[GRP], bb[GRP], sc[GRP];
#pragma unroll
                for (int j = 0; j < GRP; j++) {
                    const int cc = min(n0 + (nb * GRP + j) * 16 + 4 * g, a.N - 4);   // N % 4 == 0 (host)
                    bb[j] = *reinterpret_cast<const float4*>(bias_p + cc);
                    if constexpr (RES) rr[j] = *reinterpret_cast<const float4*>(a.R + ro + cc);
                    if constexpr (EPI == EPI_SCALE_RESADD) sc[j] = *reinterpret_cast<const float4*>(scale_p + cc);
                }
#pragma unroll
                for (int j = 0; j < GRP; j++) {
                    const int n = nb * GRP + j, col = n0 + n * 16 + 4 * g;
                    float4 v = make_float4(acc[t][n][0], acc[t][n][1], acc[t][n][2], acc[t][n][3]);
                    const float4 b = has_bias ? bb[j] : z4;
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                    if constexpr (ROPE) {   // the lane's four columns are two (even, odd) pairs of one head
                        const bool rot = col < a.rope_cols;
                        const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
                        // products and sums rounded one by one, as the reference's x*c - y*s is (rope.go:81-105): contraction is off in this
                        // epilogue.  Left to the compiler, the unrolled copies got different fused forms, and equal rows in different
                        // tile positions differed in the last bit (found by the slot-symmetry test at full size)
                        const float y0 = x0 * cs[j].x - x1 * sn[j].x, y1 = x0 * sn[j].x + x1 * cs[j].x;
                        const float y2 = x2 * cs[j].y - x3 * sn[j].y, y3 = x2 * sn[j].y + x3 * cs[j].y;
                        v.x = rot ? y0 : x0; v.y = rot ? y1 : x1; v.z = rot ? y2 : x2; v.w = rot ? y3 : x3;
                    }
                    if constexpr (EPI == EPI_GELU) { v.x = gelu1(v.x); v.y = gelu1(v.y); v.z = gelu1(v.z); v.w = gelu1(v.w); }
                    if constexpr (EPI == EPI_ELU) { v.x = elu_fast(v.x); v.y = elu_fast(v.y); v.z = elu_fast(v.z); v.w = elu_fast(v.w); }
                    if constexpr (EPI == EPI_RESADD) { v.x = rr[j].x + v.x; v.y = rr[j].y + v.y; v.z = rr[j].z + v.z; v.w = rr[j].w + v.w; }
                    if constexpr (EPI == EPI_SCALE_RESADD) {
                        const float4 s = has_scale ? sc[j] : make_float4(1.f, 1.f, 1.f, 1.f);
                        v.x = rr[j].x + s.x * v.x; v.y = rr[j].y + s.y * v.y; v.z = rr[j].z + s.z * v.z; v.w = rr[j].w + s.w * v.w;
                    }
                    if constexpr (EPI == EPI_RESADD_ELU) {
                        v.x = elu_fast(rr[j].x + v.x); v.y = elu_fast(rr[j].y + v.y); v.z = elu_fast(rr[j].z + v.z); v.w = elu_fast(rr[j].w + v.w);
                    }
                    if constexpr (!(ABL & 8)) {
                        if (m < a.M && col < a.N) *reinterpret_cast<float4*>(a.C + ro + col) = v;
                    } else {
                        if (v.x == 1.2345f && m < a.M && col < a.N) *reinterpret_cast<float4*>(a.C + ro + col) = v;
                    }
                }
            }
        }
    };
    switch (a.epi) {
        case EPI_NONE:
            if (a.rope_cos && n0 < a.rope_cols) epilogue(ic<EPI_NONE>{}, ic<1>{});   // (a column tile of v alone has nothing to rotate)
            else epilogue(ic<EPI_NONE>{}, ic<0>{});
            break;
        case EPI_GELU: epilogue(ic<EPI_GELU>{}, ic<0>{}); break;
        case EPI_ELU: epilogue(ic<EPI_ELU>{}, ic<0>{}); break;
        case EPI_RESADD: epilogue(ic<EPI_RESADD>{}, ic<0>{}); break;
        case EPI_SCALE_RESADD: epilogue(ic<EPI_SCALE_RESADD>{}, ic<0>{}); break;
        case EPI_RESADD_ELU: epilogue(ic<EPI_RESADD_ELU>{}, ic<0>{}); break;
        default: break;   // not reached: gemm5_supported
    }
}

bool gemm5_supported(const GemmArgs& a) {
    if (a.win_taps && (a.win_taps < 2 || a.win_c % 64 || a.K != a.win_taps * a.win_c || a.kslice)) return false;
    const bool res = a.epi >= EPI_RESADD;
    const bool epi_ok = a.epi == EPI_NONE || a.epi == EPI_GELU || a.epi == EPI_ELU || a.epi == EPI_RESADD || a.epi == EPI_SCALE_RESADD || a.epi == EPI_RESADD_ELU;
    constexpr int min_m = 1024;
    return a.w_bf16 && epi_ok && a.M >= min_m && a.K % 64 == 0 && a.K >= 64 && a.N % 128 == 0 && !a.tail &&
           (!a.kslice || (a.kslice % 64 == 0 && a.K % a.kslice == 0 && a.epi == EPI_NONE && !a.bias && !a.rope_cos && a.zstride % 4 == 0)) &&
           aligned16(a.A) && a.amap.ld % 4 == 0 && a.amap.batch_stride % 4 == 0 && a.ldw % 8 == 0 && aligned16(a.W) && (int64_t)a.N * a.ldw * 2 >= (int64_t)a.N * 4 &&
           aligned16(a.C) && a.cmap.ld % 4 == 0 && a.cmap.batch_stride % 4 == 0 && (!a.bias || aligned16(a.bias)) && (!a.scale || aligned16(a.scale)) &&
           (!res || aligned16(a.R)) && (!a.rope_cos || (a.rope_hd == 64 && a.rope_cols % 64 == 0 && a.epi == EPI_NONE));
}

template <int BN, int NW, int ABL = 0>
static void launch5_cfg(const GemmArgs& a, hipStream_t stream) {
    const int ncol = (a.N + BN - 1) / BN, npan = (a.M + NW * 32 - 1) / (NW * 32);
    dim3 grid((unsigned)(((npan + 7) / 8) * 8 * ncol), (unsigned)(a.kslice ? a.K / a.kslice : 1));
    if (a.aop == AOP_ELU) hipLaunchKernelGGL((k_gemm5<BN, NW, true, ABL>), grid, dim3(NW * 64), 0, stream, a);
    else hipLaunchKernelGGL((k_gemm5<BN, NW, false, ABL>), grid, dim3(NW * 64), 0, stream, a);
}

void launch_gemm5(const GemmArgs& a, hipStream_t stream) {
    note_launch(a.rope_cos ? "k_gemm5+rope" : a.kslice ? "k_gemm5+splitk" : "k_gemm5");
    const int cfg = g_gemm5_cfg;
    const bool wide = a.N % 256 == 0;   // (N = 640: 128-column tiles)
    switch (cfg) {
        case 1: launch5_cfg<256, 8>(a, stream); return;
        case 2: launch5_cfg<256, 4>(a, stream); return;
        case 3: launch5_cfg<128, 8>(a, stream); return;
        case 4: launch5_cfg<128, 4>(a, stream); return;
#ifdef PTTS_GEMM_PROBE
        case 11: launch5_cfg<256, 8, 1>(a, stream); return;
        case 12: launch5_cfg<256, 8, 2>(a, stream); return;
        case 14: launch5_cfg<256, 8, 4>(a, stream); return;
        case 18: launch5_cfg<256, 8, 8>(a, stream); return;
        case 26: launch5_cfg<256, 8, 16>(a, stream); return;
        case 17: launch5_cfg<256, 8, 7>(a, stream); return;    // MFMA + stores only
        case 25: launch5_cfg<256, 8, 15>(a, stream); return;   // MFMA only
#endif
        default: break;
    }
    if (a.M < 16384) launch5_cfg<128, 4>(a, stream);     // few row panels (the prompt prefill): 128 x 128 tiles, two blocks per CU
    else if (wide) launch5_cfg<256, 8>(a, stream);
    else launch5_cfg<128, 8>(a, stream);
}

}  // namespace ptts
