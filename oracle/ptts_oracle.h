/*
 * ptts_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the go-pocket-tts native-safetensors CPU backend
 * (internal/tts/runtime_native_safetensors.go, internal/native/<all>.go,
 * internal/runtime/{ops,tensor}/<all>.go).  Every function cites the reference
 * file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product library
 * (go-pocket-tts_amd/csrc) never links or calls it.
 *
 * Parity pinning: the reference is Go and cannot be built in this image (no Go
 * toolchain), so oracle/_ref does not exist.  The oracle is pinned against the
 * known-answer vectors the reference's own unit tests hold (transcribed as data
 * in tests/golden/reference_kat.json).  No golden tensor for the full model
 * exists in the reference (its model-level tests need the real checkpoint), so
 * model-level parity is "oracle restatement vs HIP on a synthetic checkpoint".
 */
#ifndef PTTS_ORACLE_H
#define PTTS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- worker configuration (tensor/runtime.go:18, ops/conv_runtime.go:36) ---- */
void po_set_workers(int tensor_workers, int conv_workers);
/* 1: AVX2+FMA dot/axpy order (dot_amd64.s); 0: generic order (dot.go:11-39). */
void po_set_use_avx2(int on);

/* ---- leaf primitives (runtime/tensor) ---- */
float po_dot(const float* a, const float* b, int64_t n);               /* dot.go:42, dot_amd64.go:13 */
float po_dot_generic(const float* a, const float* b, int64_t n);       /* dot.go:11-39 */
float po_dot_avx2_order(const float* a, const float* b, int64_t n);    /* dot_amd64.s:34-116 (AVX2+FMA intrinsics when built with -mavx2 -mfma) */
float po_dot_avx2_emul(const float* a, const float* b, int64_t n);     /* the same order, scalar fmaf per lane */
void  po_axpy(float* dst, int64_t ndst, float alpha, const float* src, int64_t nsrc); /* axpy.go:5-13 */
int   po_softmax_lastdim(const float* x, int64_t outer, int64_t d, float* y);  /* nn_ops.go:15-76 */
int   po_layernorm(const float* x, const float* w, const float* b, float eps,
                   int64_t outer, int64_t d, float* y);                /* linear.go:265-329, nn_ops.go:79-149 */
int   po_linear(const float* x, const float* w, const float* bias,
                int64_t batch, int64_t in, int64_t out, float* y);     /* linear.go:117-182 */
int   po_matmul2d(const float* a, const float* b, int64_t m, int64_t k, int64_t n, float* c); /* nn_ops.go MatMul */

/* ---- ops (runtime/ops) ---- */
int po_rope(float* x, const float* cos_t, const float* sin_t, int64_t prefix,
            int64_t seq, int64_t dim, int64_t pos);                    /* rope.go:81-105 (in place) */
int po_attention(const float* q, const float* k, const float* v,
                 int64_t b, int64_t h, int64_t tq, int64_t tk, int64_t d, int64_t dv,
                 int causal, int64_t offset, float* out);              /* attention.go:131-305 */
int po_attention_positions(const float* q, const float* k, const float* v,
                 int64_t b, int64_t h, int64_t tq, int64_t tk, int64_t d, int64_t dv,
                 const int64_t* posq, const int64_t* posk, int64_t context,
                 float* out);                                          /* attention.go:307-484 */
int64_t po_conv1d_outlen(int64_t len, int64_t k, int64_t stride, int64_t lpad, int64_t rpad, int64_t dil);
int po_conv1d(const float* in, const float* w, const float* bias,
              int64_t batch, int64_t in_ch, int64_t len, int64_t out_ch, int64_t k,
              int64_t stride, int64_t lpad, int64_t rpad, int64_t dil, int64_t groups,
              float* out);                                             /* conv1d.go:20-238 */
int64_t po_convtr1d_outlen(int64_t len, int64_t k, int64_t stride, int64_t pad, int64_t outpad,
                           int64_t dil, int64_t right_trim);
int po_convtr1d(const float* in, const float* w, const float* bias,
                int64_t batch, int64_t in_ch, int64_t len, int64_t out_per_group, int64_t k,
                int64_t stride, int64_t pad, int64_t outpad, int64_t dil, int64_t groups,
                int64_t right_trim, float* out);                       /* convtranspose1d.go:73-377 */
void po_repack_convtr_kernel(const float* w, int64_t in_ch, int64_t out_ch, int64_t k, float* out); /* :16-33 */
int po_mlp_silu(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                int64_t batch, int64_t in, int64_t hid, int64_t out, float* y); /* ops/mlp.go:11-32 */

/* ---- native helpers (internal/native/tensor_util.go, model.go) ---- */
void po_gelu_erf(float* x, int64_t n);           /* tensor_util.go:84-94 */
void po_silu(float* x, int64_t n);               /* tensor_util.go:73-82 */
void po_elu(float* x, int64_t n);                /* tensor_util.go:119-128 */
int  po_rmsnorm_alpha(float* x, const float* alpha, float eps, int64_t outer, int64_t d); /* :273-326 */
void po_replace_nan(float* x, int64_t n, const float* vec, int64_t d);  /* :242-271 */
/* PCM egress: internal/audio/wav_stream.go:43-54 WritePCM16Samples and :15-41 WriteWAVHeaderStreaming */
void po_pcm16(const float* s, int64_t n, int16_t* out);
void po_wav_header_streaming(uint8_t out[44]);
int  po_denorm_latent_to_bct(const float* latent, const float* std, const float* mean,
                             int64_t b, int64_t t, int64_t d, float* out); /* model.go:349-407 */
int  po_split_voice_kv(const float* cache, int64_t b, int64_t steps, int64_t heads, int64_t hd,
                       float* k, float* v);      /* flow_transformer.go:568-631 */
void po_gaussian_zero_or_passthrough(void);      /* placeholder: noise is injected by the caller (F6) */

/* ---- model ---- */
typedef struct po_tensor {
    const char*    name;
    const float*   data;   /* already decoded to f32 (store.go:339-395 done by the python reader) */
    const int64_t* shape;
    int32_t        rank;
} po_tensor;

typedef struct po_model po_model;
typedef struct po_state po_state;

po_model* po_model_create(const po_tensor* tensors, int32_t n, char* err, int32_t errlen); /* model.go:42-65 */
void      po_model_free(po_model*);
int       po_model_dims(const po_model*, int64_t* out8); /* d_model, heads, layers, ldim, flow_dim, flow_depth, mimi_dim, n_bins */

po_state* po_state_new(const po_model*);                                /* flow_transformer.go:441-449 */
/* caches[l] points at [2,1,T,H,D] f32 for layer l; offsets[l] the integral offset (flow_transformer.go:451-552) */
po_state* po_state_from_voice(const po_model*, const float* const* caches, const int64_t* steps,
                              const int64_t* offsets, char* err, int32_t errlen);
void      po_state_free(po_state*);
int64_t   po_state_offset(const po_state*, int layer);
/* copies the valid [H, offset, D] K and V of a layer out (for KV-parity checks) */
int       po_state_read_kv(const po_state*, int layer, float* k, float* v);

int po_text_embeddings(const po_model*, const int64_t* ids, int64_t n, float* out, char* err, int32_t errlen); /* conditioner.go:31-53 */
int po_prompt(const po_model*, po_state*, const float* emb, int64_t t);  /* flow_lm.go:155-187 */
/* one AR step (flow_lm.go:238-299). noise may be NULL (== zeros: temperature <= 0, flow_lm.go:395-404). */
int po_step(const po_model*, po_state*, const float* frame_in, int lsd_steps, float eos_threshold,
            const float* noise, float* frame_out, int* is_eos, float* eos_logit, float* last_hidden);
/* stateless full-sequence forward (flow_lm.go:192-233): seq [S,32], text [T,D] -> last_hidden[D], eos */
int po_flow_main(const po_model*, const float* seq, int64_t s, const float* text, int64_t t,
                 float* last_hidden, float* eos_logit);
int po_flow_direction(const po_model*, const float* c, float s, float t, const float* x, float* out); /* flow_net.go:314-356 */
int po_latent_to_mimi(const po_model*, const float* latent, int64_t t, float* out /*[512,T]*/);      /* model.go:141-319 */
int po_mimi_decode(const po_model*, const float* x /*[512,T]*/, int64_t t, float* pcm /*[1920*T]*/);  /* mimi.go:719-789 */
int64_t po_mimi_out_len(const po_model*, int64_t t);
/* staged: the decoder transformer's output (upsample + every mimiTransformerLayer, mimi.go:733-748) as [16 T, 512] rows */
int po_mimi_transformer(const po_model*, const float* x /*[512,T]*/, int64_t t, float* out /*[16*T, 512]*/);
/* test-only: overrides the attention window of every Mimi layer (DefaultMimiConfig: 250, mimi.go:32) so that a parity test can
 * show that it would notice a wrong window */
void po_debug_set_mimi_context(po_model*, int64_t context);

typedef struct po_request {
    const int64_t* tokens; int64_t n_tokens;
    float   temperature;          /* unused by the oracle: noise is injected */
    float   eos_threshold;
    int32_t max_steps;            /* resolved step budget (runtime_native_safetensors.go:61-67) */
    int32_t lsd_steps;
    int32_t frames_after_eos;
    const float* voice_emb; int64_t voice_t;       /* [Tv, D] or NULL */
    const float* const* voice_caches;              /* per layer [2,1,T,H,D] or NULL */
    const int64_t* voice_steps; const int64_t* voice_offsets;
    const float* noise;           /* [max_steps, 32] or NULL */
} po_request;

typedef struct po_result {
    float*  pcm;      int64_t n_samples;
    float*  latents;  int32_t n_frames;
    int32_t eos_step; /* -1 if none */
    float*  eos_logits; /* [n_frames]: out_eos of every step taken (flow_lm.go:262-281), for threshold-margin checks */
} po_result;

/* runtime_native_safetensors.go:52-238 */
int  po_generate(const po_model*, const po_request*, po_result*, char* err, int32_t errlen);
void po_free_result(po_result*);

#ifdef __cplusplus
}
#endif
#endif
