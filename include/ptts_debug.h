/*
 * ptts_debug.h -- test and measurement hooks of the MI355X PocketTTS path.  NOT part of the drop-in ABI (include/ptts.h) and NOT exported by
 * libptts_hip.so: these entry points are compiled from csrc/capi_hooks.cpp into libptts_hooks.so, a small library that depends on libptts_hip.so and is
 * loaded beside it by tests/, tools/ and bench.py's measurement passes.  A host of the reference links libptts_hip.so alone and can
 * neither inject faults nor run micro-benchmarks through it.  The reference's own seam has nothing of the kind: internal/tts/runtime.go:42-45.
 */
#ifndef PTTS_DEBUG_H
#define PTTS_DEBUG_H

#include "ptts.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Pieces of one Mimi decoder-transformer layer on host rows, through the kernels the decoder itself launches for them (staged parity checks of
 * mimiTransformerLayer, mimi.go:245-441).  which = PTTS_MIMI_PIECE_QKV: norm1 -> in_proj -> RoPE of q and k at positions pos0 + (row % rows_per_seg)
 * (rows_per_seg 0: pos0 + row): x [rows, mimi_dim] -> out [rows, 3 mimi_dim] (q | k | v).  which = PTTS_MIMI_PIECE_FFN: x + layer_scale_2 *
 * linear2(gelu(linear1(norm2(x)))): x [rows, mimi_dim] -> out [rows, mimi_dim]. */
#define PTTS_MIMI_PIECE_QKV 0
#define PTTS_MIMI_PIECE_FFN 1
int  ptts_mimi_layer_piece(ptts_model* m, int32_t layer, int32_t which, const float* x, int64_t rows, int32_t pos0, int32_t rows_per_seg, float* out);

/* The same with one more observation point: transformer_out (optional) receives the decoder transformer's output, i.e. the
 * [n_utt, frames * steps_per_latent, mimi_dim] rows that MimiModel.DecodeFromLatent hands to the SEANet decoder after its
 * mimiTransformerLayer loop (mimi.go:733-748) -- the staged check of a17 (window attention, RoPE, layer_scale). */
int  ptts_decode_stages(ptts_model* m, const float* latents, int32_t n_utt, int32_t frames,
                        float* pcm, float* mimi_latent, float* transformer_out);

/* Kernel micro-benchmarks (tools/microbench.py; device-resident synthetic operands, HIP-event timing; not part of the
 * drop-in path).  ptts_debug_gemm also returns max |C_variant - C_other| between the two many-row GEMM kernels. */
int ptts_debug_time_skinny(int32_t M, int32_t N, int32_t K, int32_t w_bf16, int32_t splitk, int32_t fuse_ln, int32_t iters, float* avg_us);
int ptts_debug_skinny_stamps(int32_t M, int32_t N, int32_t K, int32_t w_bf16, int32_t splitk, int32_t fuse_ln, uint64_t* out /* [max_blocks][8] */,
                             int32_t max_blocks, int32_t* n_blocks);
int ptts_debug_step_stamps(ptts_batch* b, int32_t lsd_steps, uint64_t* out /* [cap_blocks][8] */, int64_t cap_blocks, int32_t* desc /* [cap_desc][8] */,
                           int32_t cap_desc, int32_t* n_desc);
int ptts_debug_gemm(int32_t M, int32_t N, int32_t K, int32_t w_bf16, int32_t variant, int32_t epi, int32_t iters, float* avg_us,
                    float* maxdiff);

/* The AR step's linears at 128+ rows (csrc/tall.hip), stand-alone on host operands: with ln_w the rows go through k_rowprep first (x' = x + sum of the
 * `psplit` planes [psplit][M][K] + pbias -> x_out; LayerNorm(x') -> bf16 hi / lo planes), without it x is split on the host; then
 * out = epi(A W^T + bias) by k_tall (W [N][K] row-major, rounded to bf16 by the hook; epi 0 none, 1 GELU, 4 residual add of R [M][N]); splitk > 1: `out`
 * receives the planes [splitk][M][N] (plane 0 carries R + bias); out_planes: the result leaves through the bf16 hi / lo planes the next product reads
 * (hi + lo is returned). */
int ptts_debug_tall_linear(int32_t M, int32_t N, int32_t K, int32_t epi, int32_t splitk, const float* x, const float* planes, int32_t psplit, const float* pbias,
                           const float* ln_w, const float* ln_b, float eps, const float* W, const float* bias, const float* R, int32_t out_planes, float* out,
                           float* x_out);

/* name of the kernel the calling thread's last attention launch used ("k_attn_step", "k_attn_window", "k_attn_window<ragged>",
 * "k_attention"): lets a parity test assert that it exercised the kernel it means to */
const char* ptts_debug_last_attention_kernel(void);

/* Launch census of the calling thread: writes "kernel=count;..." of the launches noted since the previous call (truncated to
 * cap - 1 characters, returns the full length), clears it, and switches counting on (1) or off (0) from here on.  Calls that
 * run their launches on the calling thread (every entry point except the dispatcher's) are covered. */
int64_t ptts_debug_launch_counts(int32_t on, char* out, int64_t cap);

/* Test hook for the bounded hand-offs of k_flow_cluster (csrc/flow_cluster.hip): the model's NEXT plain-launched AR step runs the flow net's residual
 * blocks with one workgroup withholding what it should publish for block `block` (1-based; 0 clears).  Its peers' sweeps give up after their bound, the
 * launch runs to its end, and the call that contained the step fails with PTTS_ENODEVICE ("hand-off timed out"); the exchange state is cleared, the next
 * call is clean.  Nothing in the product sets it. */
int ptts_debug_flow_cluster_inject(ptts_model* m, int32_t block);


#ifdef __cplusplus
}
#endif
#endif /* PTTS_DEBUG_H */
