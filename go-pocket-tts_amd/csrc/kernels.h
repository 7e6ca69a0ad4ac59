// kernels.h -- launchers of the hand-written gfx950 kernels (kernels.hip).
// All pointers are device pointers; every launcher enqueues on `stream` and returns.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace ptts {

// Rows of a 2-D operand may live in per-utterance segments (channels-last sequences with
// zero "history" rows in front of each utterance): row r is at
//   base + (r / rows_per_batch) * batch_stride + (r % rows_per_batch) * ld      (elements)
// rows_per_batch == 0 means one flat segment (base + r * ld).
struct RowMap {
    int64_t ld = 0;
    int64_t rows_per_batch = 0;
    int64_t batch_stride = 0;
};

enum AOp : int { AOP_NONE = 0, AOP_ELU = 1 };
enum Epi : int {
    EPI_NONE = 0,        // C = acc + bias
    EPI_GELU,            // C = gelu_erf(acc + bias)                     tensor_util.go:84-94
    EPI_SILU,            // C = silu(addvec + (acc + bias))              tensor_util.go:73-82
    EPI_ELU,             // C = elu(acc + bias)                          tensor_util.go:119-128
    EPI_RESADD,          // C = R + (acc + bias)
    EPI_SCALE_RESADD,    // C = R + scale[n] * (acc + bias)              mimi.go:275-285,351-358
    EPI_GATE_RESADD,     // C = R + gate[m, n] * (acc + bias)            flow_net.go:166-171
    EPI_AXPY,            // C = R + alpha * (acc + bias)                 flow_lm.go:346-349
    EPI_RESADD_ELU,      // C = elu(R + (acc + bias))   SEANet residual sum whose every reader applies ELU first (mimi.go:146-164,752-783)
};

struct GemmArgs {
    // C[M, N] = epi( aop(A)[M, K] * W[N, K]^T )
    const float* A = nullptr; RowMap amap;
    const void*  W = nullptr; int w_bf16 = 0; int64_t ldw = 0;
    const void*  Wt = nullptr;       // fragment-ordered copy of W for the AR-step kernel (model.cpp add_tiled), optional
    int wt_i8 = 0; const float* wscale = nullptr;   // Wt holds per-row-scaled int8 (PTTS_WEIGHTS_INT8); wscale: [N] row scales
    const float* bias = nullptr;     // [N] or null
    const float* addvec = nullptr;   // [N] or null (EPI_SILU only)
    float*       C = nullptr; RowMap cmap;
    const float* R = nullptr;        // residual, addressed like C
    const float* scale = nullptr;    // [N]
    const float* gate = nullptr; int64_t ldg = 0;
    float alpha = 1.0f;
    // k_gemm3 only: interleaved-pair RoPE (rope.go:81-105) applied to the first rope_cols output columns before they are
    // stored (q and k of a qkv projection); row m sits at position rope_pos0 + m % rope_rows_per_seg (0: m)
    const float* rope_cos = nullptr; const float* rope_sin = nullptr; int rope_cols = 0, rope_hd = 64, rope_pos0 = 0, rope_rows_per_seg = 0;
    const int32_t* rope_row_pos = nullptr;   // if set: the position of row m is rope_row_pos[m] (ragged prompt rows)
    // k_gemm3 only: split-K.  kslice > 0: ceil(K / kslice) blocks share an output tile, block z multiplies k in
    // [z kslice, (z+1) kslice) and stores its raw sums at C + z * zstride (no bias, no epilogue: the consumer adds the planes)
    int kslice = 0; int64_t zstride = 0;
    // k_gemm5 only: A's rows are the overlapping windows of a causal convolution (K = win_taps x win_c, row stride win_c): walk k channel-block-major
    // (speed only: the taps' reads of one input line then follow each other; gemm5.hip)
    int win_taps = 0, win_c = 0;
    float* tail = nullptr;           // k_skinny only: the LAST column goes, as acc + bias without the epilogue, to tail[m] instead of C
    int M = 0, N = 0, K = 0;
    int aop = AOP_NONE, epi = EPI_NONE;
};
constexpr int kStepMaxRows = 256;       // rows (utterances) one AR step takes: the step kernels' grids grow by row tiles, the flow-net cluster by 12-row tiles
constexpr int kSkinnyChunkRows = 256;   // up to this many rows a GEMM on a step matrix runs as 64-row chunks of the step kernel (launch_gemm)
void launch_gemm(const GemmArgs& a, hipStream_t stream);
bool launch_gemm_rope(const GemmArgs& a, hipStream_t stream);   // a product with the RoPE epilogue (GemmArgs::rope_cos) on k_gemm3; false: the shape is not taken (the caller rotates in a second launch)
bool gemm_wres_supported(const GemmArgs& a);   // gemm_wres.hip: K, N <= 256 with the whole weight matrix resident in LDS
void launch_gemm_wres(const GemmArgs& a, hipStream_t stream);
bool gemm2_supported(const GemmArgs& a);
void launch_gemm2(const GemmArgs& a, hipStream_t stream);
bool gemm3_supported(const GemmArgs& a);   // direct-to-register activations, 256-row blocks (gemm3.hip)
void launch_gemm3(const GemmArgs& a, hipStream_t stream);
extern thread_local int g_gemm3_cfg;
bool gemm5_supported(const GemmArgs& a);   // gemm3's data movement without its conditional loads, swizzled weight image, batched epilogue (gemm5.hip)
void launch_gemm5(const GemmArgs& a, hipStream_t stream);
extern thread_local int g_gemm5_cfg;

// Weight-streaming linear for the AR step (M <= kStepMaxRows rows): C[M,N] = epi(prologue(A)[M,K] * W[N,K]^T).
// splitk > 1: raw partial sums go to partial[z][M][N] (no bias / epilogue); a consumer adds them up in a fixed order.
// The optional prologue (SkinnyFuse; needs K == row width <= 1024, splitk == 1) folds the preceding residual update
// and LayerNorm into the activation staging.
struct StepFinish;   // defined with StepState below
struct SkinnyFuse {
    // x[row] += pgate[row] * (sum_z partial[z][row] + pbias)   (partial: [psplit][M][K])
    const float* partial = nullptr; int psplit = 0; int64_t pstride = 0;
    const float* pbias = nullptr; const float* pgate = nullptr; int64_t ldpg = 0;
    float* x_out = nullptr;          // the updated rows, written once (dense [M, K]); must not alias A
    int ln = 0;                      // 1: LayerNorm the rows before the product
    const float* ln_w = nullptr; const float* ln_b = nullptr; float eps = 1e-5f;
    const float* shift = nullptr; const float* scale = nullptr; int64_t ldmod = 0;   // adaLN modulation
    float* y_out = nullptr;          // the normalised rows, written once (dense [M, K])
    // last launch of an AR step (the Euler update that produces the frame): the step's bookkeeping (k_step_finish) rides in the
    // epilogue -- the frame is appended to the utterance's latents and the slot's counters advance.  One column block only
    // (N <= 64), so every reader of the counters in this launch has read them before the one lane per row that writes them.
    const StepFinish* fin = nullptr; // device memory
    // with fin: the block also OPENS THE NEXT STEP (fin->ch): x = input_linear(frame, NaN -> bos), fx = input_proj(next noise row), x0 = that row --
    // k_step_begin's work without its launch.  The frame then goes to the utterance's latents only (C is not written: it holds x0).
    int chain = 0;
    const float* chain_noise = nullptr; int64_t chain_noise_stride = 0;   // [B][max_steps][ldim] or null (zeros)
};
bool skinny_supported(const GemmArgs& a, int splitk);
bool skinny_fuse_supported(const GemmArgs& a, const SkinnyFuse& f);
void launch_skinny(const GemmArgs& a, const SkinnyFuse& fu, int splitk, float* partial, hipStream_t stream);
extern thread_local unsigned long long* g_skinny_stamps;
// in-situ stamps of a whole AR step (ptts_debug_step_stamps): while set, every stampable launch of the step linear writes its
// blocks' 8 ticks at base + 8 * used and notes its shape
struct SkinnyStampLog {
    unsigned long long* base = nullptr; size_t cap_blocks = 0, used_blocks = 0;
    struct Desc { int32_t M, N, K, pro, nj, cg, blocks, splitk; };
    std::vector<Desc> desc;
};
extern thread_local SkinnyStampLog* g_skinny_stamp_log;
extern thread_local hipEvent_t g_skinny_ev[2];   // when set, launch_skinny times the dispatch with them (hipExtLaunchKernel)

// The AR step's big linears at 128+ rows (tall.hip): row preparation once per row, then a 64 x 64-tile product whose operands both stream through LDS.
struct PrepArgs {   // x' = x + (sum of psplit planes, in order) + pbias -> x_out; LayerNorm(x') -> bf16 hi / lo planes [M][ldy] (and f32 rows to y_out)
    const float* x = nullptr; int64_t ldx = 0;
    const float* partial = nullptr; int psplit = 0; int64_t pstride = 0; const float* pbias = nullptr;   // planes [psplit][M][D]
    float* x_out = nullptr;                                                                            // dense [M][D]; written only with `partial`
    const float* ln_w = nullptr; const float* ln_b = nullptr; float eps = 1e-5f;
    uint16_t* yh = nullptr; uint16_t* yl = nullptr; int64_t ldy = 0;
    float* y_out = nullptr;
    int M = 0, D = 0;
};
bool rowprep_supported(const PrepArgs& a);
void launch_rowprep(const PrepArgs& a, hipStream_t stream);
struct TallArgs {   // C[M,N] = epi(A[M,K] * W[N,K]^T + bias), A = ah + al (bf16 planes), W the fragment-ordered bf16 copy (Linear::wt)
    const uint16_t* ah = nullptr; const uint16_t* al = nullptr; int64_t lda = 0;
    const void* Wt = nullptr; const float* bias = nullptr;
    const float* R = nullptr; int64_t ldr = 0;                     // EPI_RESADD; with splitk > 1: added (with the bias) into plane 0
    float* C = nullptr; int64_t ldc = 0;                           // f32 result, or
    uint16_t* ch = nullptr; uint16_t* cl = nullptr; int64_t ldp = 0;   // the result as bf16 hi / lo planes (the next product's A)
    float* partial = nullptr; int64_t zstride = 0; int splitk = 1; // splitk > 1: planes [splitk][M][N]
    int M = 0, N = 0, K = 0, epi = EPI_NONE;
};
constexpr int kTallMinRows = 128;   // from this many rows the step's in_proj / linear1 / linear2 run on k_tall (below, k_skinny's one-block-per-CU shape wins)
bool tall_supported(const TallArgs& a);
void launch_tall(const TallArgs& a, hipStream_t stream);

struct LnArgs {
    const float* x = nullptr; RowMap xmap;
    const float* w = nullptr; const float* b = nullptr;  // null -> no affine (flow_net.go:228)
    float eps = 1e-5f;
    // optional adaLN modulation y = y * (1 + scale[row]) + shift[row]   (tensor_util.go:175-193)
    const float* shift = nullptr; const float* scale = nullptr; int64_t ldmod = 0;
    float* y = nullptr; int64_t ldy = 0;   // y == null: reduction prologue only
    int rows = 0, d = 0;
    // optional fused split-K reduction + residual update, applied before the normalisation (flat rows, d % 4 == 0):
    //   x[row] += pgate[row] * pscale * (sum_z partial[z][row] + pbias)       (x is updated in place)
    const float* partial = nullptr; int splitk = 0; int64_t pstride = 0;
    const float* pbias = nullptr; const float* pgate = nullptr; int64_t ldpg = 0;
    const float* pscale = nullptr;
};
void launch_layernorm(const LnArgs& a, hipStream_t stream);
// Bessel-variance "RMS" norm of the timestep embedder (tensor_util.go:273-326), in place
void launch_rmsnorm_alpha(float* x, const float* alpha, float eps, int rows, int d, hipStream_t stream);

// rows of `table` selected by ids (conditioner.go:47, shape_ops.go Gather)
void launch_embed_gather(const float* table, const int64_t* ids, int n, int d, float* out, hipStream_t stream);
// out[r, i] = isnan(in[r, i]) ? bos[i] : in[r, i]   (tensor_util.go:242-271); in_idx (optional) picks the source row
void launch_replace_nan(const float* in, const float* bos, int rows, int d, float* out, hipStream_t stream);
// elementwise helpers
void launch_silu(float* x, int64_t n, hipStream_t stream);
void launch_copy_rows(const float* src, int64_t lds, float* dst, int64_t ldd, int rows, int d, hipStream_t stream);
// out[r, c] = sin/cos timestep features: [cos(t*f) | sin(t*f)]  (flow_net.go:52-65)
void launch_timestep_features(float t, const float* freqs, int nf, float* out, hipStream_t stream);
// y[i] = 0.5 * (a[i] + b[i])   (flow_net.go:330-335)
void launch_avg2(const float* a, const float* b, float* y, int n, hipStream_t stream);

// interleaved-pair RoPE (rope.go:81-105) on rows of a [rows, ld] buffer: for each row r and head h the
// vector at x + r*ld + col0 + h*hd is rotated with the table row pos[r]
// position of row r: pos ? pos[r] : pos_base + (rows_per_seg ? r % rows_per_seg : r)
void launch_rope_rows(float* x, RowMap xmap, int col0, int heads, int hd, const int32_t* pos, int pos_base, int rows_per_seg,
                      int rows, const float* cos_t, const float* sin_t, hipStream_t stream);

// KV cache layout: [slot][head][capacity][hd]; append K/V rows taken from a qkv buffer [rows, 3*D]
void launch_kv_append(const float* qkv, int64_t ld, int d_model, int heads, int hd, const int32_t* row_slot,
                      const int32_t* row_pos, int rows, void* kcache, void* vcache, int kv_bf16, int64_t cap,
                      hipStream_t stream);

struct AttnArgs {
    // one query per (row, head)
    const float* q = nullptr; int64_t q_ld = 0; int q_col0 = 0;   // q vector at q + rowaddr(row) + q_col0 + h*hd
    int64_t q_rows_per_batch = 0, q_batch_stride = 0;             // RowMap of the q rows (0: flat, row*q_ld)
    int pos_base = 0;                                             // added to the implicit position (row % rows_per_seg)
    // keys/values: key j of (row, head) at kbase + seg(row)*k_seg_stride + h*k_head_stride + j*k_row_stride
    const void* k = nullptr; const void* v = nullptr; int kv_bf16 = 0;
    int64_t k_seg_stride = 0, k_head_stride = 0, k_row_stride = 0;
    const int32_t* row_seg = nullptr;   // null: seg = row / rows_per_seg
    int rows_per_seg = 0;
    const int32_t* row_pos = nullptr;   // query position; null: pos = row % rows_per_seg
    const int32_t* seg_len = nullptr;   // if set (AR step): pos = seg_len[seg], rope+append fused
    int context = -1;                   // keys j with pos-context < j <= pos   (attention.go:473-484)
    float* out = nullptr; int64_t out_ld = 0;  // out + rowaddr(row) + h*hd
    int64_t o_rows_per_batch = 0, o_batch_stride = 0;
    int rows = 0, heads = 0, hd = 64;
    int max_keys = 0;                   // upper bound on keys per query (sizes the LDS score buffer)
    int keys_now = 0;                   // fused step: host-side upper bound on (cache length + 1) over the rows of THIS launch; 0: unknown
    // fused RoPE + KV append for the AR step (flow_transformer.go:340-347): q,k,v read from a qkv row
    int fused_step = 0; const float* qkv = nullptr; int64_t qkv_ld = 0; int d_model = 0;
    const float* cos_t = nullptr; const float* sin_t = nullptr; int64_t cap = 0;
    const int32_t* active = nullptr;    // per segment; inactive rows write zeros and append nothing
    // fused step only: keys j < pre_len[seg] are read from a shared prefix (a device voice, [layer][head][pre_len][hd] in the
    // cache dtype) instead of the segment's own cache rows
    const void* const* pre_k = nullptr; const void* const* pre_v = nullptr; const int32_t* pre_len = nullptr; int layer = 0;
    // ragged segments (prompt prefill on the matrix cores, attn_window.hip): rows of segment s are the packed rows
    // [rag_off[s], rag_off[s+1]) at positions rag_pos0[s] + i; rows_per_seg is then the longest segment
    const int32_t* rag_off = nullptr; const int32_t* rag_pos0 = nullptr; int rag_segs = 0;
};
extern thread_local const char* g_last_attn_kernel;
// Launch census for parity tests (ptts_debug_launch_counts): while switched on for the calling thread, every launcher notes the
// kernel it picked, so a test can assert that the path it means to check is the one that ran.  Off: one thread-local load.
extern thread_local std::map<std::string, int64_t>* g_launch_census;
inline void note_launch(const char* kernel) { if (g_launch_census) (*g_launch_census)[kernel]++; }
void launch_attention(const AttnArgs& a, hipStream_t stream);   // picks k_attn_step for the fused AR step when the cache fits one burst
bool attn_step_supported(const AttnArgs& a);
void launch_attn_step(const AttnArgs& a, hipStream_t stream);
int attn_step_keys_per_round(bool kv_bf16);       // keys one load round of the block covers (32 bf16 / 16 f32)
int attn_step_rounds(int keys, bool kv_bf16);     // rounds launch_attn_step issues for a launch bounded by `keys` (1..16)
bool attn_window_supported(const AttnArgs& a);   // Mimi sliding-window attention on the f32 matrix cores
void launch_attn_window(const AttnArgs& a, hipStream_t stream);

// latent [B, T, L] -> x[b, 1+t, :] = Wp * latent + bp  (model.go:252-319), row 0 of each utterance zeroed
void launch_projector(const float* latent, int64_t lat_bstride, const float* wp, const float* bp, int b, int t, int f0, int f1,
                      int ldim, int c, float* out, hipStream_t stream);   // frames [f0, f1) -> rows 1+f of out [B][1+t][c]
// depthwise ConvTranspose1d k=2*stride, right-trimmed (convtranspose1d.go:154-202): in [B, 1+T, C] (row 0 = zeros) ->
// out rows [B][pad + T*stride][C]; w0[r][c] multiplies x[t-1], w1[r][c] multiplies x[t]
void launch_upsample_depthwise(const float* in, const float* w0, const float* w1, const float* bias, int b, int t, int f0, int f1,
                               int c, int stride, float* out, int out_pad_rows, hipStream_t stream);   // frames [f0, f1)
// voice model-state ingestion (flow_transformer.go:568-631): raw [2,1,T,H,D] f32 -> first `offset` rows of a slot's K and V cache
void launch_voice_scatter(const float* raw, int t, int heads, int hd, int offset, int slot, void* kcache, void* vcache,
                          int kv_bf16, int64_t cap, hipStream_t stream);
// copies a compact device voice (K, V as [H][offset][hd] in the cache dtype) into the first `offset` rows of several slots
void launch_voice_apply(const void* vk, const void* vv, int offset, int heads, int hd, const int32_t* slots, int n_slots,
                        void* kcache, void* vcache, int elem_bytes, int64_t cap, hipStream_t stream);
// final causal conv Cin -> 1, kernel k, ELU on the input (mimi.go:781-783): in [B][pad+T][C] channels-last
void launch_conv_final(const float* in, int in_pad_rows, const float* w /*[k*C]*/, const float* bias, int b, int t, int t0, int t1,
                       int c, int k, int elu_in /* 0: the producer already applied ELU */, float* out /*[B][T]*/, hipStream_t stream);   // samples [t0, t1) of every utterance
void launch_zero_rows(float* base, int64_t batch_stride, int b, int64_t n, hipStream_t stream);
// PCM egress: out[i] = int16(clamp(in[i], -1, 1) * 32767), product in f64, truncation, NaN -> 0 (audio/wav_stream.go:43-54); n % 8 == 0 or any n
void launch_pcm16(const float* in, int16_t* out, int64_t n, hipStream_t stream);
// the same on samples [off, off + n) of each of `rows` rows (row_stride samples apart; off, n, row_stride % 8 == 0)
void launch_pcm16_rows(const float* in, int16_t* out, int rows, int64_t row_stride, int64_t off, int64_t n, hipStream_t stream);

// One SEANet residual block (+ optionally the final conv) as a single launch, resblock.hip.  u / uo: channels-last
// [B][pad + L][C] with `pad` zero history rows per utterance; rows [t0, t1) of every utterance are produced.
// final_conv with rows: utterance b's samples [0, lim) go straight to dst (f32, or int16 through WritePCM16Samples' arithmetic) --
// dst may be page-locked host memory: the kernel's stores are the device->host transfer
struct PcmRow { void* dst; int32_t lim; int32_t s16; };
struct ResArgs {
    const float* u = nullptr; int64_t u_bs = 0; int pad = 0;
    float* uo = nullptr;                       // elu(u + block(u)), same layout as u (not written when final_conv)
    float* pcm = nullptr; int64_t pcm_bs = 0;  // final_conv: [B][L] samples
    const PcmRow* pcm_rows = nullptr;          // final_conv: if set, used instead of pcm (t0 % 4 == 0, dst 16-byte aligned)
    const void* w1 = nullptr; const void* w1_lo = nullptr; const float* b1 = nullptr;   // conv k1 (3): fragment-ordered [H][3C]
    const void* w2 = nullptr; const void* w2_lo = nullptr; const float* b2 = nullptr;   // conv k2 (1): fragment-ordered [C][H]
    const void* wf_hi = nullptr; const void* wf_lo = nullptr; const float* bf = nullptr;  // final conv as a one-column fragment-ordered matrix (hi + lo planes), bias [1]
    int B = 0, L = 0, t0 = 0, t1 = 0;
    int C = 0, H = 0, k1 = 0, k2 = 0, kf = 0, w_bf16 = 0, final_conv = 0;
    // k_resblock_up (resblock_up.hip): the transposed convolution in front of the block computed in the same kernel -- u is then not read
    // but made from xin, the previous block's rows [B][x_pad + x_L][CI] (x_L = L / up_stride), with the fragment-ordered [stride*C][2*CI]
    // weights wup (bf16) and the bias bup [C]
    int fuse_up = 0; const float* xin = nullptr; int64_t x_bs = 0; int x_pad = 0, x_L = 0, CI = 0, up_stride = 0;
    const void* wup = nullptr; const float* bup = nullptr;
};
bool resblock_supported(const ResArgs& a);
void launch_resblock(const ResArgs& a, hipStream_t stream);
bool resblock_up_supported(const ResArgs& a);
void launch_resblock_up(const ResArgs& a, hipStream_t stream);


// The feed-forward half of a Mimi decoder-transformer layer as one kernel (ffn_fused.hip): x += ls * linear2(gelu(linear1(LayerNorm(x)))),
// rows of x updated in place; img: the per-chunk weight images of model.cpp add_ffn_image (bf16)
struct FfnArgs {
    float* x = nullptr; RowMap xmap;
    const float* ln_w = nullptr; const float* ln_b = nullptr; float eps = 1e-5f;
    const void* img = nullptr;
    const float* ls = nullptr;      // [D] layer scale or null (1)
    int M = 0, D = 0, F = 0;
};
// LayerNorm + linear (+ RoPE on the first rope_cols output columns, head width 64) with register-resident rows (ffn_fused.hip k_mimi_rowlin):
// y[M][N] = rope(LayerNorm(x)[M][512] * W[N][512]^T); img: W as W1-format chunk images (model.cpp add_w1_image, bf16); no bias
struct RowLinArgs {
    const float* x = nullptr; RowMap xmap;
    const float* ln_w = nullptr; const float* ln_b = nullptr; float eps = 1e-5f;
    const void* img = nullptr;
    float* y = nullptr; RowMap ymap;
    const float* rope_cos = nullptr; const float* rope_sin = nullptr; int rope_cols = 0, rope_pos0 = 0, rope_rows_per_seg = 0;   // tables [pos][32]
    int M = 0, N = 0, K = 0;
};
bool mimi_rowlin_supported(const RowLinArgs& a);
void launch_mimi_rowlin(const RowLinArgs& a, hipStream_t stream);
bool mimi_ffn_supported(const FfnArgs& a);
void launch_mimi_ffn(const FfnArgs& a, hipStream_t stream);

// The residual blocks of the flow net (flow_net.go:116-172: `depth` x [adaLN-modulated LayerNorm -> linear + SiLU -> linear, gated, + residual], all C x C) as ONE
// launch (flow_cluster.hip): 8 workgroups per 16-row tile, each owning 64 output columns of every linear; the rows go round between the eight through tagged
// 8-byte granules.  Weights: the step kernel's fragment-ordered bf16 copies (Lin::wt).
constexpr int FC_MAX_DEPTH = 8;
struct FlowClusterArgs {
    const float* fx_in = nullptr; float* fx_out = nullptr;   // [rows][C] the residual stream in / out (may be the same buffer)
    const float* ada = nullptr; int64_t ldmod = 0;           // adaLN rows [rows][ldmod]: block r at 3 r C: shift | scale | gate
    int rows = 0, depth = 0;
    float eps[FC_MAX_DEPTH] = {};
    const float* ln_w[FC_MAX_DEPTH] = {}; const float* ln_b[FC_MAX_DEPTH] = {};
    const void* w0[FC_MAX_DEPTH] = {}; const float* b0[FC_MAX_DEPTH] = {};
    const void* w2[FC_MAX_DEPTH] = {}; const float* b2[FC_MAX_DEPTH] = {};
    unsigned long long* xbuf = nullptr;   // granules {value, tag}: [tile of 12 rows][2][16][C]
    unsigned long long* stamps = nullptr; // measurement only (null in the product): [workgroup][64] timestamps
    int inject = 0;                       // test hook (0 in the product): workgroup 7 of tile 0 publishes nothing for mlp0 of block inject - 1 -> its peers' sweeps time out
    unsigned* sync = nullptr;             // [tile] the tag base of the tile's next launch, 32 words apart; word 32 * kFlowClusterMaxTiles: fault flags
};
constexpr int kFlowClusterRows = 12;                                        // rows of the batch per tile (8 workgroups each)
constexpr int kFlowClusterMaxTiles = (kStepMaxRows + kFlowClusterRows - 1) / kFlowClusterRows;   // 22 tiles = 176 workgroups at 256 rows: one per CU, all resident at once
constexpr size_t kFlowClusterTileBytes = (size_t)2 * 16 * 512 * 8;          // a tile's two granule buffers
constexpr size_t kFlowClusterSyncBytes = (size_t)(32 * kFlowClusterMaxTiles + 32) * 4;
inline size_t flow_cluster_xbuf_bytes(int rows) { return (size_t)((rows + kFlowClusterRows - 1) / kFlowClusterRows) * kFlowClusterTileBytes; }
// whether a grid of 8 workgroups per 12-row tile is resident at once on `device` (the hand-offs between a tile's workgroups spin: they need their peers running)
bool flow_cluster_fits(int rows, int device);
bool flow_cluster_supported(const FlowClusterArgs& a, int C);
void launch_flow_cluster(const FlowClusterArgs& a, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);

// AR-step bookkeeping (runtime_native_safetensors.go:176-192 per slot, on device)
struct StepState {
    int32_t* kv_len;        // [B] keys in the cache (== flowTransformerLayerState.offset)
    int32_t* active;        // [B]
    int32_t* step;          // [B] next frame index
    int32_t* countdown;     // [B] -1 == nil
    int32_t* n_frames;      // [B]
    int32_t* eos_step;      // [B]
    int32_t* max_steps;     // [B]
    int32_t* frames_after_eos;  // [B]
    float*   eos_threshold; // [B]
    int32_t* n_active;      // [1]
    int32_t* broke;         // [B] 1 when the loop left through the countdown `break` (no StepCallback for that step)
};
// the NEXT step's opening (what k_step_begin computes), for the step's last launch to carry (k_skinny FIN + CHAIN, step_open.h)
struct StepChain {
    const void* w_in; const float* b_in; float* x; int32_t d_in;      // input_linear [d_in][ldim] row-major -> x [B][d_in]
    const void* w_pj; const float* b_pj; float* fx; int32_t d_pj;     // input_proj [d_pj][ldim] -> fx [B][d_pj]
    const float* bos; float* x0; int32_t w_bf16; int32_t ok;          // x0: [B][ldim], the next step's noise row (or zeros); ok: the shapes fit (ldim == 32, ...)
};
struct StepFinish {         // arguments of k_step_finish, resident in device memory (Batch::fin_dev)
    StepState s;
    const float* eos_logit; // [B]
    float* latents;         // [B][lat_stride]
    int64_t lat_stride;
    int32_t ldim;
    StepChain ch;
};
// prepares the step input: in32[b] = step==0 ? bos : latents[b][step-1] with NaN -> bos, x0[b] = noise of the step (or 0);
// with `lin` also x = input_linear(in32) and fx = input_proj(x0) (the two ldim-wide linears of the step, exact f32)
struct StepOpenLinears {
    const void* w_in = nullptr; const float* b_in = nullptr; int d_in = 0; float* x = nullptr;     // [d_in][ldim] row-major
    const void* w_pj = nullptr; const float* b_pj = nullptr; int d_pj = 0; float* fx = nullptr;    // [d_pj][ldim]
    int w_bf16 = 0;
};
bool step_open_mfma_ok(const StepOpenLinears& lin, int ldim);   // the matrix-core form (step_open.h) takes these shapes; it is the form a step's last launch can chain
void launch_step_begin(const StepState& s, const float* latents, int64_t lat_stride, const float* bos, const float* noise, int64_t noise_stride,
                       int ldim, int b, float* in32, float* x0, const StepOpenLinears* lin, hipStream_t stream);
// stores the decoded frame, applies EOS logic, advances kv_len/step
void launch_step_finish(const StepState& s, const float* frame, const float* eos_logit, int ldim, int b, float* latents,
                        int64_t lat_stride, hipStream_t stream);
// device draw of the sampling noise (flow_lm.go:386-408): slot b gets rows [0, spec[b].rows) of out + b * out_stride as
// N(0,1) * spec[b].sigma, a function of (spec[b].seed, row, element) only; rows == 0 leaves the slot untouched.  ldim % 4 == 0.
struct NoiseSpec { uint64_t seed; float sigma; int32_t rows; };
void launch_noise_fill(const NoiseSpec* spec_dev, int n_slots, int max_rows, float* out, int64_t out_stride, int ldim, hipStream_t stream);
// continuous batching: (re)initialise the bookkeeping of the slots a new utterance moves into; switch cancelled slots off
struct GatherTable { struct Row { int32_t src_row, dst_off, nf, T; }; Row rows[128]; };   // (2 KB of kernel arguments)
void launch_gather_frames(const GatherTable& t, int n, const float* stage, int64_t row_stride, float* dst, int ld, hipStream_t stream);   // ld % 4 == 0
void launch_readback_i32(const int32_t* a, int na, const int32_t* b, int nb, int32_t* host, hipStream_t stream);   // host: page-locked, device-visible; [a | b]
struct SlotAdmit { int32_t slot, max_steps, frames_after_eos; float eos_threshold; int32_t kv_len, pre_len; const void* pre_k; const void* pre_v; };
void launch_slot_admit(const StepState& s, int32_t* pre_len, const void** pre_k, const void** pre_v, const SlotAdmit* dev, int n, hipStream_t stream);
void launch_slot_retire(const StepState& s, const int32_t* slots_dev, int n, hipStream_t stream);
void launch_fill_i32(int32_t* p, int32_t v, int n, hipStream_t stream);
void launch_add_i32(int32_t* p, const int32_t* inc, int n, hipStream_t stream);

// [B, C, T] <-> channels-last helpers for the op-level entry points
void launch_bct_to_btc(const float* in, int b, int c, int t, float* out, int out_pad_rows, hipStream_t stream);
void launch_btc_to_bct(const float* in, int in_pad_rows, int b, int c, int t, float* out, hipStream_t stream);

}  // namespace ptts
