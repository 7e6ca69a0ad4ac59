// runtime.h -- host-side engine above the kernels: device model, batch of FlowLM states,
// prefill, AR step (hipGraph), Mimi decode, GenerateAudio loop.
#pragma once

#include <atomic>
#include <functional>

#include "kernels.h"
#include "model.h"

namespace ptts {

struct DevBuf {
    void* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void ensure(size_t bytes) {  // grow-only
        if (bytes <= n && p) return;
        release();
        if (bytes == 0) bytes = 256;
        PTTS_HIP(hipMalloc(&p, bytes));
        n = bytes;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Batch;

struct Prof {   // bench.py measurement hook (ptts_profile_*)
    bool on = false;
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    double bytes = 0, wbytes = 0;
    int64_t launches = 0;
    hipEvent_t phase[5] = {};   // setup | prefill | AR loop on the model's stream; Mimi start | end on the decoder's stream
    bool phases_on = false;     // record them (with or without the per-launch events of `on`)
    bool phases = false;
};

// page-locked staging for the small host->device uploads of one generate call (runtime.cpp: h2d)
struct UploadArena {
    char* base = nullptr;
    size_t cap = 0, off = 0;
};

struct Model {
    Desc d;
    ptts_opts opts;
    uint8_t* arena = nullptr;
    bool own_arena = false;
    int device = 0;
    hipStream_t stream = nullptr;    // AR loop / prefill (high priority: latency-bound launches)
    hipStream_t stream2 = nullptr;   // Mimi decode of finished frame ranges, concurrent with the AR loop
    std::vector<hipEvent_t> events;
    std::mutex mu;
    std::map<int, std::unique_ptr<DevBuf>> tcomb;  // lsd_steps -> [n][flow_dim]: 0.5*(embed_s(i/n) + embed_t((i+1)/n))
    std::vector<std::unique_ptr<DevBuf>> ws;       // grow-only workspaces (Mimi decode, prefill)
    std::unique_ptr<Batch> cached_batch;
    Prof prof;
    int fc_inject = 0;   // test hook: the next k_flow_cluster launch (plain launches) runs with FlowClusterArgs::inject = this, once
    // k_flow_cluster's bounded hand-offs gave up (a tile's workgroups were not running together: a masked or shared device): the steps concerned were
    // re-issued as the 2 x depth launches (same bits) -- fc_fallbacks counts the events -- and this engine's batches keep the launches from then on
    std::atomic<int64_t> fc_fallbacks{0};
    std::atomic<bool> fc_disabled{false};

    ~Model();
    template <class T> const T* at(size_t off) const { return off == NONE ? nullptr : reinterpret_cast<const T*>(arena + off); }
    UploadArena upload;
    DevBuf& work(size_t i, size_t bytes) {
        while (ws.size() <= i) ws.emplace_back(new DevBuf());
        ws[i]->ensure(bytes);
        return *ws[i];
    }
    void use_device() const { PTTS_HIP(hipSetDevice(device)); }
    // seeds of requests that name none (noise_seed == 0): a splitmix64 stream seeded with the clock when the model is opened
    // (the reference seeds its generator the same way, runtime_native_safetensors.go:27-32); never 0
    uint64_t noise_state = 0;
    uint64_t next_noise_seed() {
        uint64_t z = (noise_state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return z ? z : 1;
    }
    const float* tcomb_for(int lsd_steps);
    void compute_tcomb(float s, float t, float* dst /* device [flow_dim] */);
};

// n_slots independent FlowLMState (flow_lm.go:45-49) in HBM + the per-step workspace
struct Batch {
    Model* m = nullptr;
    int B = 0;
    int cap = 0;             // KV capacity per slot (keys)
    int max_steps = 0;       // rows of `latents` per slot
    DevBuf kcache, vcache;   // [L][B][H][cap][hd]
    DevBuf state_i32;        // StepState arrays
    DevBuf state_f32;
    StepState st{};
    std::vector<int32_t> kv_len_host;
    // shared voice prefix: slots that took their first keys from a device-resident voice read them from the voice's own copy
    // in the AR step (one copy for the whole batch stays in L2) instead of from their private cache rows
    DevBuf pre_k, pre_v, pre_len;          // [B] device pointers to layer 0 of the voice's K / V, [B] prefix length (0: none)
    std::vector<const void*> pre_k_host, pre_v_host;
    std::vector<int32_t> pre_len_host;
    // step workspace
    DevBuf in_raw, in32, x, xn, qkv, attn, ff, last, eos, sy, ada, fx, fh, fh2, cur, noise_step, partial;
    DevBuf latents;          // [B][max_steps][ldim]
    DevBuf noise;            // [B][max_steps][ldim] or empty
    bool has_noise = false;
    bool opened = false;     // x and fx of the COMING step are in place: set by step_open and by a step whose last launch chained the next one's opening
    bool chain_ok = false;   // the shapes allow that chaining (StepFinish::ch in fin_dev)
    // a chained launch runs several blocks per row tile, so what it writes for the next step (fx, x0) must not be what its other blocks still read:
    // fx / cur alternate with fx2 / cur2 from one chained step to the next (par: which pair the coming step reads; 0 after every k_step_begin)
    DevBuf fx2, cur2;
    // the flow net's residual blocks as one launch (flow_cluster.hip): granule buffers and the tag / fault words; fc_ok: the model's shapes take it
    DevBuf fc_xbuf, fc_sync, fc_stamps;   // (fc_stamps: PTTS_FC_STAMPS measurement runs only)
    bool fc_ok = false;
    // tall.hip (batches of kTallMinRows rows and more): bf16 hi + lo row planes of the normalised residual rows [2][B][d_model] and of gelu(linear1) [2][B][ffn]
    DevBuf tp_a, tp_f;
    bool tall_ok = false;
    bool tail_fused = false;   // the step's last transformer launch also produced sy (silu(t + cond_embed)): the flow part can be re-issued from `last` / `sy` / `cur` as they stand
    unsigned* fc_fault() const { return fc_sync.as<unsigned>() + 32 * kFlowClusterMaxTiles; }
    int par = 0;
    float* fx_now() const { return (par ? fx2 : fx).as<float>(); }
    float* cur_now() const { return (par ? cur2 : cur).as<float>(); }
    hipStream_t io_stream = nullptr;   // continuous batch: voice ingestion and prefill of newcomers run here, beside the step chain on the model's stream
    bool slot_local = false; // continuous batch: per-slot device state (kv_len, voice prefix) is written slot by slot by the admit kernel, never as whole arrays
    // host-side upper bound on the cache length of any slot (set by voice/prompt ingestion, +1 per step): lets a step's
    // attention launch issue only the key loads that can be live (AttnArgs::keys_now).  capturing: a graph is being recorded,
    // its launches must cover the whole cache
    int kv_bound = 0;
    bool capturing = false;
    // graph replay of the AR step (use_graph): one captured step per attention round count (attn_step_rounds: the step
    // attention's load rounds follow the cache length, and a captured launch cannot change), captured on first use
    // [with sampling noise][attention rounds][0: one step, 1: graph_steps[.][0] steps (the short graphs), 2: graph_steps[.][1] steps (the long ones)]
    hipGraphExec_t graphs[2][17][3] = {};
    int graph_steps[2][2] = {{0, 0}, {0, 0}};   // steps per graph the second / third column was captured with
    int graph_lsd = 0;
    int capture_keys = 0;    // while capturing: the cache-length bound the recorded attention launches must cover
    // page-locked scratch of the generate loop: [0] live-utterance count, [1, 1+B) n_frames, [1+B, 1+2B) eos_step read back in
    // one copy after the loop; rows_pinned: the B result rows uploaded to the decoder (no pageable staging, no extra sync)
    int32_t* n_active_pinned = nullptr;
    PcmRow* rows_pinned = nullptr;
    DevBuf fin_dev;          // StepFinish: what the step's last launch needs for the bookkeeping it carries

    ~Batch();
    size_t kv_elem() const { return m->opts.kv == PTTS_KV_BF16 ? 2 : 4; }
    void* kc(int layer) const { return (char*)kcache.p + (size_t)layer * B * m->d.heads * cap * m->d.hd * kv_elem(); }
    void* vc(int layer) const { return (char*)vcache.p + (size_t)layer * B * m->d.heads * cap * m->d.hd * kv_elem(); }
};

// Result buffers (PCM) are page-locked so that the device-to-host copy runs at link speed (pageable: 12 GB/s measured, 5 ms per
// 64 x 10 s batch; pinned: ~50 GB/s).  Pinning is slow, so freed buffers go back to a process-wide pool.
void* result_alloc(size_t bytes);
void result_free(void* p);   // accepts pool blocks and plain malloc'ed pointers
bool result_is_pinned(const void* p);   // false for the pageable fallback blocks: those must not be written by a kernel

// a voice model state resident in HBM: K and V per layer as [H][offset][hd] in the cache dtype
struct Voice {
    Model* m = nullptr;
    int offset = 0;
    DevBuf k, v;
    size_t layer_bytes() const { return (size_t)m->d.heads * offset * m->d.hd * (m->opts.kv == PTTS_KV_BF16 ? 2 : 4); }
};
Voice* voice_create(Model& m, const float* const* caches, const int64_t* steps, const int64_t* offsets);
void batch_apply_voice(Batch& b, const Voice& v, const std::vector<int32_t>& slots);
bool voice_usable_by(const Voice& v, const Model& m);   // same GPU and cache geometry (e.g. engines made by model_share)

// buffers of one Mimi decode (all channels-last, spanning the whole utterance so that frame ranges can be decoded in order)
struct MimiWs {
    int B = 0, T = 0, P0 = 0;
    int Ls[4] = {0, 0, 0, 0}, Ps[4] = {0, 0, 0, 0};
    float *xp = nullptr, *up = nullptr, *n1 = nullptr, *qkv[MAX_LAYERS] = {nullptr}, *attn = nullptr, *ff = nullptr, *c0 = nullptr;
    float *u[3] = {nullptr, nullptr, nullptr}, *uo[3] = {nullptr, nullptr, nullptr}, *h[3] = {nullptr, nullptr, nullptr};
    bool zeroed = false;
};
constexpr int kMimiGroup = 64;   // utterances decoded together through one decoder workspace (generate: larger batches go group after group)
void mimi_setup(Model& m, MimiWs& w, int B, int T);
// pcm_rows (device array of B PcmRow, whole range only): the fused final block writes every utterance's samples straight to its
// row (page-locked host memory) instead of pcm; *rows_used says whether that path was taken (false: pcm holds the samples)
void mimi_range(Model& m, MimiWs& w, const float* lat, int64_t lat_bstride, int f0, int f1, float* pcm, float* mimi_latent, hipStream_t s,
                const PcmRow* pcm_rows = nullptr, bool* rows_used = nullptr, float* xformer_out = nullptr);

void mimi_layer_qkv(Model& m, int layer, const float* x, RowMap xmap, int R, float* qkv, RowMap qmap, int pos0, int rows_per_seg, float* n1, hipStream_t s);
void mimi_layer_ffn(Model& m, int layer, float* x, RowMap xmap, int R, float* n1, float* ffb, hipStream_t s);
Model* model_open(Plan* plan, void* device_arena, int fill);
Batch* batch_new(Model& m, int n_slots, int cap, int max_steps);
void batch_reset(Batch& b);
void batch_set_voice(Batch& b, int slot, const float* const* caches, const int64_t* steps, const int64_t* offsets);
// rows: device [R, d_model]; row_offsets host [B+1]
void batch_prompt(Batch& b, const float* rows_dev, const int64_t* row_offsets);
// core of one AR step on device state: in32 [B, ldim], cur [B, ldim] (= x0) -> cur (= frame), eos, last; appends KV at kv_len
// opened: x and fx were produced by step_open; fuse_finish: the bookkeeping of k_step_finish rides in the last launch -- returns
// whether it did (false: the caller launches k_step_finish)
// chain (with fuse_finish): the last launch also opens the next step (sets b.opened); the frame then lives in the latents only, b.cur holds the next x0
bool step_core(Batch& b, int lsd_steps, bool opened = false, bool fuse_finish = false, bool chain = false);
void step_open(Batch& b);
// A hand-off inside k_flow_cluster timed out (the fault word of the batch is set).  flow_cluster_recover clears the exchange state, switches the batch (and
// the engine's later batches) to the 2 x depth launches and counts the event; flow_cluster_fault does that and throws FlowClusterFault: the caller owns a
// retry -- generate() runs the chunk again on the launches, the staged step re-issues its flow part, the dispatcher re-queues what was in flight.
struct FlowClusterFault : Error { using Error::Error; };
void flow_cluster_recover(Batch& b);
void flow_cluster_fault(Batch& b);
void step_flow_again(Batch& b, int lsd_steps);   // the LSD decode of the step just taken, once more, from the intact `last` / `sy` and the caller-restored `cur` (staged API)
                                    // first launch of a generate step (input, noise, the two 32-wide linears)
// xformer_out (optional, staged parity checks): the decoder transformer's output rows [B][T * up_stride][mimi_dim]
void mimi_decode(Model& m, const float* lat_dev, int64_t lat_bstride, int B, int T, float* pcm_dev, float* mimi_latent_dev, float* xformer_out = nullptr);
// pieces of the generate loop that the continuous batch (continuous.cpp) reuses
struct UploadScope { UploadScope(UploadArena& a, hipStream_t s); UploadScope(UploadArena& a, hipEvent_t last_use); ~UploadScope(); };   // small uploads of this thread are staged page-locked, no waits
void h2d(void* dst, const void* src, size_t bytes, hipStream_t s);
void d2h(void* dst, const void* src, size_t bytes, hipStream_t s);
int resolve_max_steps(const ptts_request& r);                                   // runtime_native_safetensors.go:61-67
void enqueue_step(Batch& b, int lsd, bool use_graph, int nsteps = 1);           // nsteps > 1 only with use_graph
void mimi_zero_history(Model& m, MimiWs& w, hipStream_t s);
Model* model_share(Model& base);   // another engine over base's weight arena (base must outlive it)
Model* model_replicate(Model& base, int device);   // the model on another GPU of this process: own arena, copied from base's by hipMemcpyPeer
void generate(Model& m, const ptts_request* reqs, int n, ptts_result* res);
std::string request_error(const Desc& d, const ptts_request& q);   // empty: the request is well formed


// text front end (text.cpp; internal/text/prepare.go, chunk.go)
struct TextChunk {
    std::string text;
    std::vector<int64_t> token_ids;
    int num_words = 0, max_frames = 0, frames_after_eos = 0;
};
typedef std::function<std::vector<int64_t>(const std::string&)> TextEncodeFn;
int text_count_words(const std::string& s);
int text_estimate_max_frames(int64_t token_count, double frame_rate);
int text_frames_after_eos(int64_t num_words);
std::string text_prepare(const std::string& input);
std::vector<std::string> text_split_sentences(const std::string& text);
std::vector<TextChunk> text_chunks(const std::string& input, const TextEncodeFn& encode, int max_tokens, double frame_rate);

// SentencePiece unigram encoder (tokenizer.cpp; internal/tokenizer/sentencepiece.go, sentencepiece_bytes_wasm.go)
struct Tokenizer;
Tokenizer* tokenizer_from_bytes(const void* data, size_t len);
Tokenizer* tokenizer_from_path(const std::string& path);
std::vector<int64_t> tokenizer_encode(const Tokenizer& t, const std::string& text);
size_t tokenizer_vocab(const Tokenizer& t);
void tokenizer_free(Tokenizer* t);
std::string nfkc_utf8(const std::string& s);

// voice files (voicefile.cpp; internal/safetensors/reader.go:69-140,232-308, internal/native/flow_transformer.go:451-590): host only
struct VoiceFile {
    StFile st;
    int kind = 0;                                        // PTTS_VOICE_FILE_*
    std::vector<float> emb; std::vector<int64_t> emb_shape;   // LoadVoiceEmbedding: [1, T, D]
    std::string emb_error;                               // why the first tensor is no embedding (1-D, 4-D ...)
    struct Tensor { std::vector<int64_t> shape; std::vector<float> data; };
    struct Module { std::string name; std::map<std::string, Tensor> tensors; };
    std::vector<Module> modules;                         // LoadVoiceModelState: sorted by name; "current_end" already turned into "offset"
    std::string state_error;
};
VoiceFile* voice_file_from_path(const std::string& path);
VoiceFile* voice_file_from_bytes(const void* data, size_t len);
void voice_file_embedding(const VoiceFile& v, const float** data, int64_t shape[3]);
void voice_file_require_state(const VoiceFile& v);
void voice_file_state(const VoiceFile& v, int n_layers, int heads, int head_dim, const float** caches, int64_t* steps, int64_t* offsets);

// optional post-processing of a finished utterance (dsp.cpp; internal/audio/dsp.go)
void dsp_peak_normalize(float* s, int64_t n);
void dsp_dc_block(float* s, int64_t n, int sample_rate);
void dsp_fade_in(float* s, int64_t n, int sample_rate, double ms);
void dsp_fade_out(float* s, int64_t n, int sample_rate, double ms);

// the weight broadcast of a multi-GPU start-up (broadcast.cpp)
void rccl_unique_id(uint8_t out[128]);
void rccl_broadcast(void* device_buf, size_t bytes, int rank, int n_ranks, const uint8_t id[128], int device);

// continuous batch of one model (continuous.cpp): fixed slots / KV capacity / step budget; used by the dispatcher's continuous mode
struct ContEngine;
ContEngine* cont_create(Model& m, int slots, int kv_cap, int max_steps);
void cont_destroy(ContEngine* e);
bool cont_accepts(const ContEngine& e, const ptts_request& r);   // fits the geometry and needs no per-step host work
int cont_free_slots(const ContEngine& e);
int cont_busy(const ContEngine& e);                               // utterances generating + groups being decoded
bool cont_admit_now(const ContEngine& e, int waiting);              // admission pacing: see continuous.cpp
void cont_admit(ContEngine& e, const ptts_request* const* reqs, ptts_result* const* results, void* const* tags, int n);
void cont_advance(ContEngine& e, int steps, std::vector<void*>& done, bool drain);
void cont_occupancy(const ContEngine& e, int64_t* steps, int64_t* slot_steps);   // AR steps launched so far; utterances stepping in them, summed
void cont_abort(ContEngine& e, int code, std::vector<void*>& done);

// request dispatcher (dispatcher.cpp)
struct Dispatcher;
typedef int (*ExecFn)(void* user, int worker, const ptts_request* reqs, int32_t n, ptts_result* results, char* err, int32_t errlen);
struct DispatchCont { int on = 0, kv_capacity = 0, max_steps = 0, steps_per_group = 0; };   // ptts_dispatch_opts: continuous batching
Dispatcher* dispatcher_create(Model* const* models, int n_models, ExecFn exec, void* user, int n_workers, int max_batch, int window_us, int queue_cap,
                              const DispatchCont* cont = nullptr);
void dispatcher_close(Dispatcher* d);
int dispatcher_generate(Dispatcher* d, const ptts_request* req, ptts_result* res, std::string* err);
void dispatcher_stats(Dispatcher* d, ptts_dispatch_stats* out);

}  // namespace ptts
