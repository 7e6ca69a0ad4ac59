/*
 * ptts.h -- C ABI of libptts_hip.so: the MI355X-native PocketTTS synthesis path.
 *
 * Drop-in boundary.  The reference has no FFI; its seam is the Go interface
 *     type Runtime interface { GenerateAudio(ctx, tokens []int64, cfg RuntimeGenerateConfig) ([]float32, error); Close() }
 * (internal/tts/runtime.go:42-45) implemented by nativeSafetensorsRuntime
 * (internal/tts/runtime_native_safetensors.go:20-244) on top of native.Model
 * (internal/native/model.go:25-138).  The entry points below are exactly what a
 * cgo binding of that seam needs; each one names the reference method it
 * replaces.  INTEGRATION.md shows the cgo shim.
 *
 * Conventions (mirroring the reference's):
 *   - every call returns 0 on success or a PTTS_E* code; the message is
 *     ptts_last_error() (thread-local), worded like the reference's Go errors;
 *   - no exceptions cross the ABI; inputs are borrowed and never mutated;
 *   - outputs are library-allocated and released with ptts_free_result(), or
 *     caller-allocated where a size is known up front;
 *   - a model handle may be shared by threads; ptts_generate() calls on one
 *     model are serialised on that model's HIP stream.
 *   - there is NO CPU fallback: without a usable HIP device every compute
 *     entry point fails with PTTS_ENODEVICE.
 */
#ifndef PTTS_H
#define PTTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTTS_OK          0
#define PTTS_EINVAL      1   /* bad argument / shape / range (reference: fmt.Errorf paths) */
#define PTTS_EIO         2   /* file cannot be read */
#define PTTS_EFORMAT     3   /* not a valid safetensors / missing tensor */
#define PTTS_ENODEVICE   4   /* no HIP device, or a HIP call failed */
#define PTTS_ECANCELLED  5   /* ctx.Err() between steps (runtime_native_safetensors.go:156-159) */
#define PTTS_ENOMEM      6

/* how weights are held in HBM */
#define PTTS_WEIGHTS_F32   0  /* as the reference holds them (every dtype decoded to f32, store.go:339-395) */
#define PTTS_WEIGHTS_BF16  1  /* 2 bytes/param; exact when the file itself is BF16 */
#define PTTS_WEIGHTS_INT8  2  /* weight-only int8 for every matrix the AR step streams (1 byte/param, per-row f32 scale: q = rint(W / s),
                               * s = max|row| / 127; the effective weights q*s are used consistently by prefill and step); the rest as
                               * PTTS_WEIGHTS_BF16.  Not a reference mode (the reference's int8 is ONNX quantize_dynamic,
                               * scripts/export_onnx.py:319-331): tolerance stated against the f32 oracle in tests/test_gpu_int8.py */
/* KV-cache element type */
#define PTTS_KV_F32   0
#define PTTS_KV_BF16  1

typedef struct ptts_model ptts_model;
typedef struct ptts_plan  ptts_plan;
typedef struct ptts_batch ptts_batch;
typedef struct ptts_voice ptts_voice;

typedef struct ptts_opts {
    int32_t device;          /* HIP device ordinal */
    int32_t weights;         /* PTTS_WEIGHTS_* */
    int32_t kv;              /* PTTS_KV_* */
    int32_t max_batch;       /* utterances stepped together on the GPU (default 64, at most 256) */
    int32_t use_graph;       /* 0 (default): the ~47 launches of an AR step are issued per step -- the host stays ~3x ahead of
                                the GPU and there is no gap between steps; 1: the step is captured once into a hipGraph and
                                replayed (one host call per step, ~8 us of idle GPU between replays: up to 3 % slower, but the
                                launching thread needs a fraction of the CPU time) */
    int32_t reserved0;       /* must be 0 (round 2's step_plan: the alternative launch plans were measurements, not options; they live in tools/probes/step_plans) */
    int32_t reserved[10];
} ptts_opts;

void ptts_default_opts(ptts_opts* o);

typedef struct ptts_info {
    int64_t d_model, n_heads, n_layers, ffn, ldim, n_bins;      /* flow_lm.go:13-27 */
    int64_t flow_dim, flow_depth;                               /* flow_net.go:242-248 */
    int64_t mimi_dim, mimi_heads, mimi_layers, mimi_context;    /* mimi.go:16-34 */
    int64_t sample_rate, samples_per_frame, steps_per_latent;   /* MimiTiming(): runtime_native_safetensors.go:40-49 */
    double  frame_rate, encoder_frame_rate;
    int64_t n_params, arena_bytes;
    int32_t weights, kv;
} ptts_info;

/* ---- model lifetime: native.LoadModelFromSafetensors / LoadModelFromStore / Model.Close
 *      (internal/native/model.go:33-71) ---- */
int  ptts_model_open(const char* safetensors_path, const ptts_opts* opts, ptts_model** out);
int  ptts_model_open_bytes(const void* data, size_t len, const ptts_opts* opts, ptts_model** out);
void ptts_model_close(ptts_model* m);                               /* nil-safe, like Model.Close */
int  ptts_model_info(const ptts_model* m, ptts_info* out);
const char* ptts_last_error(void);

/* Two-phase open for multi-GPU start-up: every rank plans from the
 * file header alone, allocates one device arena of ptts_plan_arena_bytes(), rank 0
 * fills it (fill=1: decode, convert, derive tables, upload) and the arena is then
 * broadcast once over RCCL/xGMI; the other ranks adopt it as is (fill=0). */
int    ptts_plan_create(const char* safetensors_path, const ptts_opts* opts, ptts_plan** out);
int    ptts_plan_create_bytes(const void* data, size_t len, const ptts_opts* opts, ptts_plan** out);
size_t ptts_plan_arena_bytes(const ptts_plan* p);
int    ptts_model_open_planned(ptts_plan* p, void* device_arena, int fill, ptts_model** out); /* consumes p */
/* The broadcast itself, for hosts that have no collective library of their own (the reference's Go server): rank 0 makes the id
 * (ncclGetUniqueId), hands the 128 bytes to the other processes over the channel that started them, then EVERY rank calls
 * ptts_rccl_broadcast on its arena (root = rank 0; one ncclCommInitRank + ncclBroadcast + ncclCommDestroy over RCCL / xGMI).
 * librccl is loaded on first use (PTTS_RCCL_LIB overrides the name).
 * STATUS: exercised with n_ranks == 1 only (tests/test_gpu_model.py); the two-rank test (tests/test_gpu_multirank.py) needs a box with
 * two GPUs and has not run.  Compare a checksum of the arena across ranks after the hand-over (bench.py open_model does). */
int    ptts_rccl_unique_id(uint8_t out[128]);
int    ptts_rccl_broadcast(void* device_buf, size_t bytes, int32_t rank, int32_t n_ranks, const uint8_t id[128], int32_t device);
/* host image of the arena (ptts_plan_arena_bytes() bytes), for hosts that upload / broadcast it themselves; needs no GPU */
int    ptts_plan_fill_host(const ptts_plan* p, void* host_arena);
void   ptts_plan_free(ptts_plan* p);

/* ---- the Runtime seam: tts.Runtime.GenerateAudio (runtime_native_safetensors.go:52-238) ---- */
typedef void (*ptts_step_callback)(void* user, int32_t step, int32_t max_steps); /* RuntimeGenerateConfig.StepCallback */

typedef void (*ptts_pcm_callback)(void* user, int64_t sample_offset, int64_t n_samples, const void* samples);

#define PTTS_PCM_F32 0
#define PTTS_PCM_S16 1

typedef struct ptts_request {
    const int64_t* tokens; int64_t n_tokens;          /* must be non-empty (:57-59) */
    float   temperature;                              /* RuntimeGenerateConfig.Temperature: sampling noise = N(0,1) * sqrt(max(t, 0)), drawn on the
                                                       * device per (noise_seed, step) when `noise` is NULL; <= 0: zero noise (flow_lm.go:386-408) */
    float   eos_threshold;                            /* isEOS = logit > threshold (flow_lm.go:281) */
    int32_t max_steps;                                /* <=0: estimated_max_steps, then EstimateMaxFrames (:61-67) */
    int32_t estimated_max_steps;
    int32_t lsd_steps;                                /* <=0 -> 1 (:69-72) */
    int32_t frames_after_eos;                         /* text.FramesAfterEOS */
    /* voice conditioning, mutually exclusive (:100-102) */
    const float* voice_embedding; int64_t voice_frames;        /* [Tv, d_model], prepended (:104-119) */
    const float* const* voice_caches;                           /* per layer [2,1,T,H,Dh] f32 (flow_transformer.go:451-552) */
    const int64_t* voice_cache_steps;                           /* per layer T */
    const int64_t* voice_offsets;                               /* per layer offset, 0 <= offset <= T */
    /* injected sampling noise (reproducible runs, parity tests): [noise_rows, ldim] = N(0,1)*sqrt(temperature) draws, one row
     * per AR step, consumed as x0 of that step's LSD decode (flow_lm.go:283-288); replaces the device draw.  NULL: the library
     * draws (temperature > 0) or uses zeros (temperature <= 0). */
    const float* noise;
    ptts_step_callback step_callback; void* callback_user;
    const volatile int32_t* cancel;                   /* polled between steps; nonzero -> PTTS_ECANCELLED */
    int32_t want_latents;                             /* 1: also return the latent frames */
    int32_t pcm_format;                               /* PTTS_PCM_F32 (0): result.pcm; PTTS_PCM_S16 (1): result.pcm16, encoded on the device */
    /* a voice model state already resident in HBM (ptts_voice_create); exclusive with the two host forms above.
     * The reference loads the voice file once per Synthesize call and rebuilds the FlowLM state from it for
     * every chunk (service.go:127,216-246, flow_lm.go:134-145); the device copy is that cached voice. */
    const ptts_voice* voice;
    uint64_t noise_seed;                              /* device draw: the stream of this request; 0 = a fresh one per request from the model's
                                                       * own generator, seeded with the clock at open like the reference's (runtime_native_safetensors.go:27-32) */
    int32_t noise_rows;                               /* rows behind `noise`; must cover the resolved step budget (0: not checked, the caller vouches) */
    int32_t reserved[1];
    /* Frame-granular streaming (the /tts/stream path, server.go:354-396, at finer grain than the reference's per-chunk
     * PCMChunk): finished frame ranges are decoded while the AR loop is still running and handed over in order, each sample
     * exactly once, from a library thread; `samples` points into the buffer the result will own (float or int16 per
     * pcm_format).  The callback must not call into this library.  stream_frames: frames per hand-over (<= 0: 12 = 0.96 s). */
    ptts_pcm_callback pcm_callback; void* pcm_user;
    int32_t stream_frames;
    int32_t reserved2[3];
} ptts_request;

typedef struct ptts_result {
    float*  pcm;       int64_t n_samples;             /* fresh copy owned by the caller (:237) */
    float*  latents;   int32_t n_frames;              /* [n_frames, ldim] when want_latents */
    int32_t eos_step;                                 /* first step whose EOS logit crossed the threshold, -1 if none */
    int32_t status;                                   /* per-request PTTS_* code */
    int16_t* pcm16;                                   /* PTTS_PCM_S16: n_samples little-endian samples, v = int16(clamp(s, -1, 1) * 32767)
                                                       * exactly as audio.WritePCM16Samples (internal/audio/wav_stream.go:43-54); pcm is NULL then */
    int32_t reserved[2];
} ptts_result;

/* A second ENGINE over the weights of `base` (its own streams, KV caches, workspaces; the weight arena is shared and
 * read-only): two engines on one GPU, e.g. behind one dispatcher, let one batch's Mimi decode run beside the next batch's
 * prefill + AR loop.  `base` must outlive the engine; a device voice belongs to the engine it was uploaded to. */
int  ptts_model_share(ptts_model* base, ptts_model** out);
/* The model on ANOTHER GPU of the same process (one process, N GPUs: the shape of the reference's single server with its N workers,
 * internal/server/server.go:119-143,398-421): a private weight arena on `device`, copied from base's over the GPUs' direct link (hipMemcpyPeer) -- the file
 * is read and decoded once, no collective library is involved.  Independent of `base` afterwards (either may be closed first).  A dispatcher over the N
 * models (ptts_dispatcher_create) deals requests to them; a device voice belongs to the GPU it was uploaded to.  `device` may be base's own. */
int  ptts_model_replicate(ptts_model* base, int32_t device, ptts_model** out);
/* ptts_opts.use_graph of an open model, changed between calls (A/B measurement, hosts that become short of CPU) */
int  ptts_model_set_use_graph(ptts_model* m, int32_t use_graph);
/* ptts_opts.max_batch of an open model or engine (1..256), changed between calls: how many utterances of one ptts_generate call are stepped
 * together (the reference's counterpart is its worker count, internal/server/server.go:132-134, internal/config/config.go:87); the engine's
 * KV caches and workspaces follow on the next call */
int  ptts_model_set_max_batch(ptts_model* m, int32_t max_batch);

/* n_reqs == 1 reproduces GenerateAudio exactly.  n_reqs > 1 is this library's batching
 * extension: independent utterance chunks stepped together (per-row EOS countdown, ragged KV). */
int  ptts_generate(ptts_model* m, const ptts_request* reqs, int32_t n_reqs, ptts_result* results);
void ptts_free_result(ptts_result* r);

/* The 44-byte header audio.WriteWAVHeaderStreaming emits in front of a PCM16 stream (internal/audio/wav_stream.go:15-41):
 * 24 kHz, mono, 16 bit, RIFF and data sizes 0xFFFFFFFF. */
void ptts_wav_header_streaming(uint8_t out[44]);

/* Optional post-processing of a finished utterance, in place, in the order the CLI applies it (cmd/pockettts/synth.go:361-390):
 * PeakNormalize, DCBlock (20 Hz high-pass), FadeIn, FadeOut (internal/audio/dsp.go:12-78); 24 kHz.  Host samples, host code.
 * Normalise and the fades are bit-exact restatements; the DC block's biquad comes from a third-party module in the reference and
 * is held to the properties the reference's tests state (parity unpinned). */
int  ptts_dsp_apply(float* samples, int64_t n, int32_t normalize, int32_t dc_block, double fade_in_ms, double fade_out_ms);

/* ---- Text front end (internal/text/prepare.go, chunk.go) -------------------------------------------------
 * What Synthesize does before it calls the runtime: normalise the text, cut it into sentence-based chunks of <= max_tokens
 * tokens, and derive each chunk's step budget and EOS tail.  The SentencePiece encoder is the caller's. */
int32_t ptts_text_estimate_max_frames(int64_t token_count, double frame_rate);   /* EstimateMaxFrames, prepare.go:38-48 */
int32_t ptts_text_frames_after_eos(int64_t num_words);                            /* FramesAfterEOS, prepare.go:53-59 */
/* PrepareText (prepare.go:66-100).  Writes up to cap bytes (no terminator) and the full length to *out_len. */
int  ptts_text_prepare(const char* utf8, int64_t len, char* out, int64_t cap, int64_t* out_len);
/* Encoder callback: writes up to cap ids and returns the count (> cap: called again with room), < 0: error. */
typedef int64_t (*ptts_encode_fn)(void* user, const char* utf8, int64_t len, int64_t* ids, int64_t cap);

/* SentencePiece unigram encoder (tokenizer.NewSentencePieceTokenizer / Tokenizer.Encode, internal/tokenizer/sentencepiece.go:19-40;
 * algorithm: internal/tokenizer/sentencepiece_bytes_wasm.go = go-sentencepiece-encoder v1.1.1): reads the pieces of a
 * SentencePiece ModelProto (`tokenizer.model`), normalises (control characters dropped, White_Space -> ' ', NFKC), prepends and
 * substitutes U+2581, Viterbi over the piece trie, consecutive unknowns merged. */
typedef struct ptts_tokenizer ptts_tokenizer;
int     ptts_tokenizer_open(const char* model_path, ptts_tokenizer** out);
int     ptts_tokenizer_open_bytes(const void* data, size_t len, ptts_tokenizer** out);
void    ptts_tokenizer_free(ptts_tokenizer* t);
int64_t ptts_tokenizer_vocab_size(const ptts_tokenizer* t);
/* Encode: writes up to cap ids, returns the count (> cap: call again with room; an empty text gives 0), < 0 on error */
int64_t ptts_tokenizer_encode(const ptts_tokenizer* t, const char* utf8, int64_t len, int64_t* ids, int64_t cap);
/* the same with the signature of the encoder callback, user = the ptts_tokenizer: ptts_text_chunks(text, len, ptts_tokenizer_encode_cb, tok, ...);
 * ptts_text_chunks also takes encode == NULL with user = a ptts_tokenizer as "use the built-in encoder" */
int64_t ptts_tokenizer_encode_cb(void* user, const char* utf8, int64_t len, int64_t* ids, int64_t cap);
/* NFKC as the tokenizer applies it (generated Unicode tables) */
int     ptts_text_nfkc(const char* utf8, int64_t len, char* out, int64_t cap, int64_t* out_len);
typedef struct ptts_chunks ptts_chunks;
typedef struct ptts_chunk_info {
    const char* text; int64_t text_len;            /* PrepareText of the joined sentences */
    const int64_t* token_ids; int64_t n_tokens;
    int32_t num_words;                             /* of the raw sentences (prepare.go:140) */
    int32_t max_frames;                            /* EstimateMaxFrames(n_tokens, frame_rate) */
    int32_t frames_after_eos;
    int32_t reserved;
} ptts_chunk_info;
/* PrepareChunks (prepare.go:105-184); max_tokens: 50 in the reference's service; frame_rate <= 0: 12.5 */
int  ptts_text_chunks(const char* utf8, int64_t len, ptts_encode_fn encode, void* user, int32_t max_tokens, double frame_rate, ptts_chunks** out);
int32_t ptts_chunks_count(const ptts_chunks* c);
int  ptts_chunks_get(const ptts_chunks* c, int32_t i, ptts_chunk_info* out);
void ptts_chunks_free(ptts_chunks* c);

/* ---- Request dispatcher (what the reference's worker pool becomes) ----------------------------------
 * internal/server/server.go:132-134,398-421 admits `workers` concurrent Synthesize calls through a semaphore; here callers
 * block in ptts_dispatch_generate and a worker thread per model coalesces waiting requests (up to max_batch, for at most
 * window_us after the oldest one arrived) into one batched generate.  A request cancelled while it waits is answered
 * PTTS_ECANCELLED without running.  Several models (one per GPU) pull from the same queue; nothing is exchanged between them.
 * A request that carries a device voice is served by the model that voice was uploaded to. */
typedef struct ptts_dispatcher ptts_dispatcher;
typedef struct ptts_dispatch_opts {
    int32_t max_batch;    /* <= 0: the model's max_batch */
    int32_t window_us;    /* coalescing window, counted from the arrival of the oldest waiting request; stretched (to at most
                           * 4 windows) while requests keep arriving less than window_us / 4 apart */
    int32_t queue_cap;    /* <= 0: 4096; a full queue answers PTTS_ENOMEM */
    int32_t continuous;   /* 1: continuous batching -- one long-lived batch per model: between groups of AR steps, utterances that have
                           * ended leave for the decoder and waiting requests take their slots (their prompts prefilled as one ragged
                           * launch).  Requests with a step / PCM callback, lsd_steps > 1 or budgets beyond the two limits below run
                           * batch-at-a-time while the engine is empty.  -1: batch-at-a-time for everything.  0 (the default): continuous
                           * when every model of the dispatcher has its GPU to itself (where it wins uniform AND mixed-length traffic),
                           * batch-at-a-time when two of them share one (ptts_model_share: the two-engine setting) */
    int32_t cont_kv_capacity;      /* keys per slot (voice prefix + prompt + steps), <= 0: 512 (the reach of the one-burst step attention with a bf16 cache) */
    int32_t cont_max_steps;        /* step budget per utterance, <= 0: 256 (EstimateMaxFrames of a 50-token chunk is 234) */
    int32_t cont_steps_per_group;  /* AR steps between two looks at the slots, <= 0: 3 */
    int32_t reserved[1];
} ptts_dispatch_opts;
typedef struct ptts_dispatch_stats {
    int64_t requests, batches, cancelled_waiting, max_queue_depth;
    double  mean_batch, mean_wait_us, mean_exec_us;
    int64_t cont_steps, cont_slot_steps;   /* continuous batching: AR steps launched; utterances stepping in them, summed (ratio = mean occupied slots) */
    int64_t flow_cluster_fallbacks;        /* times a hand-off inside k_flow_cluster timed out on one of the dispatcher's models: the steps concerned were redone as
                                            * launches (same bits, nobody failed) and that engine keeps the launches from then on */
} ptts_dispatch_stats;
int  ptts_dispatcher_create(ptts_model* const* models, int32_t n_models, const ptts_dispatch_opts* opts, ptts_dispatcher** out);
/* queueing logic over a caller-supplied executor (no GPU involved): unit tests of the coalescing / cancellation rules */
typedef int (*ptts_dispatch_exec)(void* user, int32_t worker, const ptts_request* reqs, int32_t n, ptts_result* results, char* err, int32_t errlen);
int  ptts_dispatcher_create_custom(ptts_dispatch_exec exec, void* user, int32_t n_workers, const ptts_dispatch_opts* opts, ptts_dispatcher** out);
int  ptts_dispatch_generate(ptts_dispatcher* d, const ptts_request* req, ptts_result* result);   /* blocks until the request has run */
void ptts_dispatcher_stats(ptts_dispatcher* d, ptts_dispatch_stats* out);
void ptts_dispatcher_close(ptts_dispatcher* d);   /* waits for queued work; later calls are refused */

/* Uploads a voice model state ([2,1,T,H,Dh] f32 per layer + offsets; safetensors.LoadVoiceModelState,
 * reader.go:127-140,273-308) once; requests then reference it by handle. */
int  ptts_voice_create(ptts_model* m, const float* const* caches, const int64_t* cache_steps, const int64_t* offsets, ptts_voice** out);
void ptts_voice_free(ptts_voice* v);

/* ---- voice FILES (internal/safetensors/reader.go:69-155,219-308).  The reference's Service reads the voice path of a request with
 *      InspectVoiceFile and then either LoadVoiceModelState or LoadVoiceEmbedding (internal/tts/service.go:216-246); these entry points
 *      are those three functions plus the consumer-side checks of flowTransformer.initStateFromVoiceModelState
 *      (internal/native/flow_transformer.go:451-590).  Host only: no GPU is touched until ptts_voice_open uploads. ---- */
#define PTTS_VOICE_FILE_UNKNOWN      0   /* VoiceFileUnknown */
#define PTTS_VOICE_FILE_EMBEDDING    1   /* VoiceFileEmbedding: legacy `audio_prompt` (or any first tensor) [T, D] / [1, T, D] */
#define PTTS_VOICE_FILE_MODEL_STATE  2   /* VoiceFileModelState: `<module>/cache` [2,B,T,H,D] + `<module>/offset` (or legacy `<module>/current_end`) */
typedef struct ptts_voice_file ptts_voice_file;
int  ptts_voice_file_open(const char* path, ptts_voice_file** out);                        /* OpenStore + classifyVoiceTensorNames + load */
int  ptts_voice_file_open_bytes(const void* data, size_t len, ptts_voice_file** out);      /* ...FromBytes; the bytes are copied */
void ptts_voice_file_close(ptts_voice_file* f);
int32_t ptts_voice_file_kind(const ptts_voice_file* f);                                    /* InspectVoiceFile: PTTS_VOICE_FILE_* */
/* LoadVoiceEmbedding (reader.go:69-85,219-230): the first tensor (sorted names) as [1, T, D]; *data stays owned by f and is what a
 * request carries as voice_embedding (voice_frames = shape[1]; shape[2] must be the model's d_model).  PTTS_EFORMAT with the
 * reference's message for a model-state file and for a tensor that is not 2-D or 3-D. */
int  ptts_voice_file_embedding(const ptts_voice_file* f, const float** data, int64_t shape[3]);
/* LoadVoiceModelState (reader.go:127-140,273-308) as the reference's map of modules: count, then per module its name and tensors
 * ("cache", "offset"; a legacy `current_end` has already become offset = [float(len(current_end))], shape [1]).  PTTS_EFORMAT with the
 * reference's message for an embedding file or a tensor name without "<module>/<key>". */
typedef struct ptts_voice_tensor { const float* data; int64_t count; int32_t rank; int32_t reserved; int64_t shape[8]; } ptts_voice_tensor;
int  ptts_voice_file_modules(const ptts_voice_file* f, int32_t* n_modules);
int  ptts_voice_file_module(const ptts_voice_file* f, int32_t i, const char** name, ptts_voice_tensor* cache /* data NULL: absent */, ptts_voice_tensor* offset);
/* initStateFromVoiceModelState (flow_transformer.go:451-480,517-590) up to the re-layout: for layers 0..n_layers-1 the module
 * "transformer.layers.{i}.self_attn" must hold a cache [2,1,T,heads,head_dim] and an integral offset 0 <= offset <= T (heads /
 * head_dim 0: unchecked, as a layer without them is in the reference).  Fills what ptts_request.voice_caches / voice_cache_steps /
 * voice_offsets and ptts_voice_create take (pointers owned by f).  PTTS_EINVAL with the reference's messages. */
int  ptts_voice_file_state(const ptts_voice_file* f, int32_t n_layers, int32_t heads, int32_t head_dim,
                           const float** caches, int64_t* cache_steps, int64_t* offsets);
/* a model-state voice file straight into HBM: ptts_voice_file_open[_bytes] + ptts_voice_file_state (the model's dimensions) +
 * ptts_voice_create.  An embedding file answers PTTS_EFORMAT (LoadVoiceModelState's message): read it with ptts_voice_file_embedding. */
int  ptts_voice_open(ptts_model* m, const char* path, ptts_voice** out);
int  ptts_voice_open_bytes(ptts_model* m, const void* data, size_t len, ptts_voice** out);

/* Measurement hook for bench.py: while enabled the AR step runs eagerly (no graph) with a HIP event pair around
 * every launch of the dominant (weight-streaming linear) kernel on the model's stream. */
typedef struct ptts_profile {
    int64_t launches;
    double  total_ms;            /* sum of the event-pair durations */
    double  algorithmic_bytes;   /* sum over launches of weights + activations in + out */
    char    kernel[64];
    double  weight_bytes;        /* the weights-only part of algorithmic_bytes (each step linear's matrix once per launch) */
    double  prefill_ms, ar_loop_ms, mimi_ms;   /* device time of the phases of the last ptts_generate call (HIP events on its streams) */
} ptts_profile;
int ptts_profile_enable(ptts_model* m, int32_t on);   /* 0: off; 1: per-launch events (plain launches) + phase times; 2: phase times only (the call runs as configured) */
int ptts_profile_read(ptts_model* m, ptts_profile* out);   /* returns and resets the counters */

/* ---- staged entry points = the native.Model methods GenerateAudio calls (model.go:76-138,141,410).
 *      A ptts_batch is n_slots independent FlowLMState objects (flow_lm.go:45-49) held in HBM. ---- */
int  ptts_text_embeddings(ptts_model* m, const int64_t* ids, int64_t n, float* out /* [n, d_model] host */); /* Model.TextEmbeddings */
int  ptts_batch_new(ptts_model* m, int32_t n_slots, int32_t kv_capacity, ptts_batch** out);                 /* Model.NewFlowState x n_slots */
void ptts_batch_free(ptts_batch* b);
int  ptts_batch_reset(ptts_batch* b);
/* Model.NewFlowStateFromVoiceModelState for one slot */
int  ptts_batch_set_voice_state(ptts_batch* b, int32_t slot, const float* const* caches,
                                const int64_t* cache_steps, const int64_t* offsets);
/* Model.PromptFlow for every slot at once: slot s gets emb + row_offsets[s] .. row_offsets[s+1] (rows of d_model floats, host) */
int  ptts_batch_prompt(ptts_batch* b, const float* emb, const int64_t* row_offsets);
/* Model.SampleNextLatentStateful for every slot: frames_in [n_slots, ldim] (NaN = BOS), noise NULL or [n_slots, ldim];
 * outputs (host, caller-allocated, any may be NULL): frames_out [n_slots, ldim], eos_logits [n_slots], last_hidden [n_slots, d_model] */
int  ptts_batch_step(ptts_batch* b, const float* frames_in, int32_t lsd_steps, const float* noise,
                     float* frames_out, float* eos_logits, float* last_hidden);
int  ptts_batch_offsets(ptts_batch* b, int64_t* out /* [n_slots] */);
int  ptts_batch_read_kv(ptts_batch* b, int32_t slot, int32_t layer, float* k, float* v /* [H, offset, Dh] each */);
/* Model.LatentToMimi + Model.MimiDecode: latents [n_utt, frames, ldim] host -> pcm [n_utt, frames*samples_per_frame] host;
 * mimi_latent (optional) receives LatentToMimi's [n_utt, mimi_dim, frames] */
int  ptts_decode_latents(ptts_model* m, const float* latents, int32_t n_utt, int32_t frames,
                         float* pcm, float* mimi_latent);
/* Voice cloning, the part the reference holds natively: projectSpeakerConditioning
 * (internal/onnx/voice_encode.go:119-158) -- Mimi-encoder latents [frames, 512] (host) times flow_lm.speaker_proj_weight
 * [d_model, 512] -> voice embedding [frames, d_model] (host), which a request then carries as `voice_embedding`.  The Mimi encoder
 * itself has no native reference (mimi.go:14,791-794: ErrMimiEncoderNotImplemented) and is not built.  PTTS_EFORMAT when the
 * checkpoint has no speaker projection tensor. */
int  ptts_speaker_project(ptts_model* m, const float* latent, int64_t frames, float* out);
/* The device draw of FlowLM.makeGaussianNoise (flow_lm.go:386-408) for one request: out[rows, ldim] (host) receives exactly the
 * rows ptts_generate would consume for (noise_seed = seed, temperature) -- so that a test can hand the same noise to a reference. */
int  ptts_noise_rows(ptts_model* m, uint64_t seed, float temperature, int32_t rows, float* out);
/* FlowLM.FlowDirection (flow_lm.go:302-308): c [n, d_model], x [n, ldim] -> out [n, ldim] */
int  ptts_flow_direction(ptts_model* m, const float* c, float s, float t, const float* x, int32_t n, float* out);

/* ---- kernel-level entry points (host buffers in/out) = internal/runtime/ops + tensor, backed by
 *      the same HIP kernels the model path launches; they exist so that the reference's
 *      known-answer tests can be replayed against the GPU kernels. ---- */
int ptts_op_linear(const float* x, const float* w, const float* bias, int64_t rows, int64_t in, int64_t out, float* y);              /* tensor.Linear nn_ops.go:268 */
int ptts_op_layernorm(const float* x, const float* w, const float* b, float eps, int64_t rows, int64_t d, float* y);                 /* tensor.LayerNorm nn_ops.go:79 */
int ptts_op_rope(float* x /* [prefix, seq, dim] in place */, const float* cos_t, const float* sin_t, int64_t table_rows,
                 int64_t prefix, int64_t seq, int64_t dim, int64_t pos);                                                              /* ops.RoPE rope.go:13 */
int ptts_op_attention_positions(const float* q, const float* k, const float* v, int64_t b, int64_t h, int64_t tq, int64_t tk,
                 int64_t d, const int64_t* posq, const int64_t* posk, int64_t context, float* out);                                  /* ops.AttentionWithPositions attention.go:63 */
int ptts_op_conv1d_leftpad(const float* x /* [B,Cin,L] */, const float* w /* [Cout,Cin,k] */, const float* bias,
                 int64_t b, int64_t cin, int64_t len, int64_t cout, int64_t k, float* y /* [B,Cout,L] */);                           /* ops.Conv1DLeftPad conv1d.go:95 (stride 1, leftPad k-1) */
int ptts_op_convtr1d_righttrim(const float* x /* [B,Cin,L] */, const float* w /* [Cin,Cout/groups,k] */, const float* bias,
                 int64_t b, int64_t cin, int64_t len, int64_t cout_per_group, int64_t k, int64_t stride, int64_t groups,
                 float* y /* [B,Cout,L*stride] */);                                                                                  /* ops.ConvTranspose1DRightTrim convtranspose1d.go:213 (k = 2*stride, trim k-stride; groups 1 or Cin) */

/* audio.WritePCM16Samples on the device (internal/audio/wav_stream.go:43-54), without the byte packing */
int ptts_op_pcm16(const float* samples, int64_t n, int16_t* out);

/* build/version string, e.g. "ptts-hip 0.2 gfx950" */
const char* ptts_version(void);

/* Test and measurement hooks (launch census, in-kernel stamps, micro-benchmarks, staged observation points of the decoder, the fault injection of
 * k_flow_cluster's hand-offs) are NOT in this library: they are declared in ptts_debug.h and live in libptts_hooks.so, which the tests load beside it. */

#ifdef __cplusplus
}
#endif
#endif /* PTTS_H */
