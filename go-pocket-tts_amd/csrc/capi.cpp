// capi.cpp -- extern "C" surface of libptts_hip.so (include/ptts.h).
#include <cmath>
#include <map>

#include "runtime.h"

#include "capi_internal.h"

using namespace ptts;
using namespace ptts::capi;

extern "C" {

void ptts_default_opts(ptts_opts* o) {
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->device = 0;
    o->weights = PTTS_WEIGHTS_F32;
    o->kv = PTTS_KV_F32;
    o->max_batch = 64;
    o->use_graph = 0;
}

const char* ptts_last_error(void) { return last_error_ref().c_str(); }
const char* ptts_version(void) { return "ptts-hip 0.2 gfx950"; }

int ptts_plan_create(const char* path, const ptts_opts* opts, ptts_plan** out) {
    return guard([&] {
        if (!path || !out) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        std::unique_ptr<ptts_plan> pl(new ptts_plan());
        pl->p.opts = resolve_opts(opts);
        st_open_path(path, pl->p.file);
        plan_build(pl->p);
        *out = pl.release();
    });
}

int ptts_plan_create_bytes(const void* data, size_t len, const ptts_opts* opts, ptts_plan** out) {
    return guard([&] {
        if (!data || !out) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        std::unique_ptr<ptts_plan> pl(new ptts_plan());
        pl->p.opts = resolve_opts(opts);
        pl->p.file.owned.assign((const uint8_t*)data, (const uint8_t*)data + len);  // OpenStoreFromBytes keeps the bytes
        pl->p.file.data = pl->p.file.owned.data();
        pl->p.file.size = len;
        st_parse(pl->p.file);
        plan_build(pl->p);
        *out = pl.release();
    });
}

size_t ptts_plan_arena_bytes(const ptts_plan* p) { return p ? p->p.desc.total_bytes : 0; }

int ptts_plan_fill_host(const ptts_plan* p, void* host_arena) {
    return guard([&] {
        if (!p || !host_arena) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        std::memset(host_arena, 0, p->p.desc.total_bytes);
        plan_fill(p->p, reinterpret_cast<uint8_t*>(host_arena));
    });
}
void ptts_plan_free(ptts_plan* p) { delete p; }

int ptts_dsp_apply(float* samples, int64_t n, int32_t normalize, int32_t dc_block, double fade_in_ms, double fade_out_ms) {
    return guard([&] {
        if ((!samples && n > 0) || n < 0) throw Error(PTTS_EINVAL, "audio: nil samples");
        const int sr = 24000;   // audio.ExpectedSampleRate
        if (normalize) dsp_peak_normalize(samples, n);
        if (dc_block) dsp_dc_block(samples, n, sr);
        if (fade_in_ms > 0) dsp_fade_in(samples, n, sr, fade_in_ms);
        if (fade_out_ms > 0) dsp_fade_out(samples, n, sr, fade_out_ms);
    });
}

int ptts_rccl_unique_id(uint8_t out[128]) {
    return guard([&] {
        if (!out) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        require_device();
        rccl_unique_id(out);
    });
}

int ptts_rccl_broadcast(void* device_buf, size_t bytes, int32_t rank, int32_t n_ranks, const uint8_t id[128], int32_t device) {
    return guard([&] {
        require_device();
        rccl_broadcast(device_buf, bytes, rank, n_ranks, id, device);
    });
}

int ptts_model_open_planned(ptts_plan* p, void* device_arena, int fill, ptts_model** out) {
    int rc = guard([&] {
        if (!p || !out) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        std::unique_ptr<ptts_model> h(new ptts_model());
        h->m = model_open(&p->p, device_arena, fill);
        *out = h.release();
    });
    delete p;
    return rc;
}

int ptts_model_share(ptts_model* base, ptts_model** out) {
    return guard([&] {
        if (!base || !base->m || !out) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        std::unique_ptr<ptts_model> h(new ptts_model());
        h->m = model_share(*base->m);
        *out = h.release();
    });
}

int ptts_model_replicate(ptts_model* base, int32_t device, ptts_model** out) {
    return guard([&] {
        if (!base || !base->m || !out) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        std::unique_ptr<ptts_model> h(new ptts_model());
        h->m = model_replicate(*base->m, device);
        *out = h.release();
    });
}

int ptts_model_set_use_graph(ptts_model* m, int32_t use_graph) {
    return guard([&] {
        if (!m || !m->m) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        std::lock_guard<std::mutex> lock(m->m->mu);   // not under a running generate
        m->m->opts.use_graph = use_graph ? 1 : 0;
    });
}

int ptts_model_set_max_batch(ptts_model* m, int32_t max_batch) {
    return guard([&] {
        if (!m || !m->m) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        if (max_batch < 1 || max_batch > kStepMaxRows) throw Error(PTTS_EINVAL, strfmt("ptts-hip: max_batch %d outside [1, %d]", max_batch, kStepMaxRows));
        std::lock_guard<std::mutex> lock(m->m->mu);   // not under a running generate
        m->m->opts.max_batch = max_batch;
    });
}

int ptts_model_open(const char* path, const ptts_opts* opts, ptts_model** out) {
    ptts_plan* pl = nullptr;
    int rc = ptts_plan_create(path, opts, &pl);
    if (rc) return rc;
    return ptts_model_open_planned(pl, nullptr, 1, out);
}

int ptts_model_open_bytes(const void* data, size_t len, const ptts_opts* opts, ptts_model** out) {
    ptts_plan* pl = nullptr;
    int rc = ptts_plan_create_bytes(data, len, opts, &pl);
    if (rc) return rc;
    return ptts_model_open_planned(pl, nullptr, 1, out);
}

void ptts_model_close(ptts_model* m) {
    if (!m) return;
    if (m->m) {
        (void)hipSetDevice(m->m->device);
        delete m->m;
    }
    delete m;
}

int ptts_model_info(const ptts_model* h, ptts_info* o) {
    return guard([&] {
        if (!h || !h->m || !o) throw Error(PTTS_EINVAL, "native-safetensors runtime unavailable");
        const Desc& d = h->m->d;
        std::memset(o, 0, sizeof *o);
        o->d_model = d.d_model; o->n_heads = d.heads; o->n_layers = d.n_layers; o->ffn = d.ffn; o->ldim = d.ldim; o->n_bins = d.n_bins;
        o->flow_dim = d.flow_dim; o->flow_depth = d.flow_depth;
        o->mimi_dim = d.mimi_dim; o->mimi_heads = d.mimi_heads; o->mimi_layers = d.mimi_layers; o->mimi_context = d.mimi_ctx;
        o->sample_rate = 24000; o->samples_per_frame = d.samples_per_frame; o->steps_per_latent = d.up_stride;
        o->frame_rate = 12.5; o->encoder_frame_rate = 200.0;  // mimi.go:25-34
        o->n_params = d.n_params; o->arena_bytes = (int64_t)d.total_bytes;
        o->weights = h->m->opts.weights; o->kv = h->m->opts.kv;
    });
}

int ptts_generate(ptts_model* h, const ptts_request* reqs, int32_t n, ptts_result* results) {
    return guard([&] {
        if (!h || !h->m) throw Error(PTTS_EINVAL, "native-safetensors runtime unavailable");
        if (!reqs || !results || n <= 0) throw Error(PTTS_EINVAL, "ptts-hip: no requests");
        set_last_error("");
        generate(*h->m, reqs, n, results);
        for (int i = 0; i < n; i++)
            if (results[i].status != PTTS_OK) {
                std::string msg = last_error_ref();
                throw Error(results[i].status, msg.empty() ? "generate: request failed" : msg);
            }
    });
}

void ptts_free_result(ptts_result* r) {
    if (!r) return;
    result_free(r->pcm);
    result_free(r->pcm16);
    free(r->latents);
    r->pcm = nullptr;
    r->pcm16 = nullptr;
    r->latents = nullptr;
    r->n_samples = 0;
    r->n_frames = 0;
}

struct ptts_chunks { std::vector<TextChunk> v; };

int32_t ptts_text_estimate_max_frames(int64_t token_count, double frame_rate) { return text_estimate_max_frames(token_count, frame_rate); }
int32_t ptts_text_frames_after_eos(int64_t num_words) { return text_frames_after_eos(num_words); }

int ptts_text_prepare(const char* utf8, int64_t len, char* out, int64_t cap, int64_t* out_len) {
    return guard([&] {
        if ((!utf8 && len > 0) || len < 0 || !out_len) throw Error(PTTS_EINVAL, "text: nil argument");
        const std::string r = text_prepare(std::string(utf8 ? utf8 : "", (size_t)len));
        *out_len = (int64_t)r.size();
        if (out && cap > 0) memcpy(out, r.data(), (size_t)std::min<int64_t>(cap, (int64_t)r.size()));
    });
}

struct ptts_tokenizer { Tokenizer* t; };

int ptts_tokenizer_open(const char* model_path, ptts_tokenizer** out) {
    return guard([&] {
        if (!out) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        std::unique_ptr<ptts_tokenizer> h(new ptts_tokenizer());
        h->t = tokenizer_from_path(model_path ? model_path : "");
        *out = h.release();
    });
}

int ptts_tokenizer_open_bytes(const void* data, size_t len, ptts_tokenizer** out) {
    return guard([&] {
        if (!out) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        std::unique_ptr<ptts_tokenizer> h(new ptts_tokenizer());
        h->t = tokenizer_from_bytes(data, len);
        *out = h.release();
    });
}

void ptts_tokenizer_free(ptts_tokenizer* t) {
    if (!t) return;
    tokenizer_free(t->t);
    delete t;
}

int64_t ptts_tokenizer_vocab_size(const ptts_tokenizer* t) { return t && t->t ? (int64_t)tokenizer_vocab(*t->t) : 0; }

int64_t ptts_tokenizer_encode(const ptts_tokenizer* t, const char* utf8, int64_t len, int64_t* ids, int64_t cap) {
    int64_t n = -1;
    guard([&] {
        if (!t || !t->t) throw Error(PTTS_EINVAL, "tokenizer is nil");
        if ((!utf8 && len > 0) || len < 0) throw Error(PTTS_EINVAL, "tokenizer: nil text");
        const std::vector<int64_t> v = tokenizer_encode(*t->t, std::string(utf8 ? utf8 : "", (size_t)len));
        for (int64_t i = 0; i < (int64_t)v.size() && i < cap && ids; i++) ids[i] = v[(size_t)i];
        n = (int64_t)v.size();
    });
    return n;
}

int64_t ptts_tokenizer_encode_cb(void* user, const char* utf8, int64_t len, int64_t* ids, int64_t cap) {
    return ptts_tokenizer_encode(reinterpret_cast<const ptts_tokenizer*>(user), utf8, len, ids, cap);
}

int ptts_text_nfkc(const char* utf8, int64_t len, char* out, int64_t cap, int64_t* out_len) {
    return guard([&] {
        if ((!utf8 && len > 0) || len < 0 || !out_len) throw Error(PTTS_EINVAL, "text: nil argument");
        const std::string r = nfkc_utf8(std::string(utf8 ? utf8 : "", (size_t)len));
        *out_len = (int64_t)r.size();
        if (out && cap > 0) std::memcpy(out, r.data(), (size_t)std::min<int64_t>(cap, (int64_t)r.size()));
    });
}

int ptts_text_chunks(const char* utf8, int64_t len, ptts_encode_fn encode, void* user, int32_t max_tokens, double frame_rate, ptts_chunks** out) {
    return guard([&] {
        if (!encode && user) encode = ptts_tokenizer_encode_cb;   // default encoder: `user` is a ptts_tokenizer
        if ((!utf8 && len > 0) || len < 0 || !encode || !out) throw Error(PTTS_EINVAL, "text: nil argument");
        TextEncodeFn enc = [&](const std::string& t) {
            std::vector<int64_t> ids(64);
            int64_t n = encode(user, t.data(), (int64_t)t.size(), ids.data(), (int64_t)ids.size());
            if (n > (int64_t)ids.size()) {
                ids.resize((size_t)n);
                n = encode(user, t.data(), (int64_t)t.size(), ids.data(), (int64_t)ids.size());
            }
            if (n < 0) throw Error(PTTS_EINVAL, "encode \"" + t + "\": tokenizer error");
            ids.resize((size_t)n);
            return ids;
        };
        std::unique_ptr<ptts_chunks> c(new ptts_chunks());
        c->v = text_chunks(std::string(utf8 ? utf8 : "", (size_t)len), enc, max_tokens, frame_rate);
        *out = c.release();
    });
}

int32_t ptts_chunks_count(const ptts_chunks* c) { return c ? (int32_t)c->v.size() : 0; }

int ptts_chunks_get(const ptts_chunks* c, int32_t i, ptts_chunk_info* out) {
    return guard([&] {
        if (!c || !out || i < 0 || i >= (int32_t)c->v.size()) throw Error(PTTS_EINVAL, "text: chunk index out of range");
        const TextChunk& t = c->v[(size_t)i];
        std::memset(out, 0, sizeof *out);
        out->text = t.text.data(); out->text_len = (int64_t)t.text.size();
        out->token_ids = t.token_ids.data(); out->n_tokens = (int64_t)t.token_ids.size();
        out->num_words = t.num_words; out->max_frames = t.max_frames; out->frames_after_eos = t.frames_after_eos;
    });
}

void ptts_chunks_free(ptts_chunks* c) { delete c; }

struct ptts_dispatcher { Dispatcher* d; };

int ptts_dispatcher_create(ptts_model* const* models, int32_t n_models, const ptts_dispatch_opts* o, ptts_dispatcher** out) {
    return guard([&] {
        if (!models || n_models <= 0 || !out) throw Error(PTTS_EINVAL, "dispatcher: at least one model is required");
        std::vector<Model*> ms;
        for (int i = 0; i < n_models; i++) {
            if (!models[i] || !models[i]->m) throw Error(PTTS_EINVAL, "dispatcher: nil model");
            ms.push_back(models[i]->m);
        }
        DispatchCont dc;
        if (o) { dc.on = o->continuous; dc.kv_capacity = o->cont_kv_capacity; dc.max_steps = o->cont_max_steps; dc.steps_per_group = o->cont_steps_per_group; }
        *out = new ptts_dispatcher{dispatcher_create(ms.data(), n_models, nullptr, nullptr, 0, o ? o->max_batch : 0, o ? o->window_us : 2000, o ? o->queue_cap : 0, &dc)};
    });
}

int ptts_dispatcher_create_custom(ptts_dispatch_exec exec, void* user, int32_t n_workers, const ptts_dispatch_opts* o, ptts_dispatcher** out) {
    return guard([&] {
        if (!exec || !out) throw Error(PTTS_EINVAL, "dispatcher: nil executor");
        *out = new ptts_dispatcher{dispatcher_create(nullptr, 0, exec, user, n_workers, o ? o->max_batch : 0, o ? o->window_us : 2000, o ? o->queue_cap : 0)};
    });
}

int ptts_dispatch_generate(ptts_dispatcher* d, const ptts_request* req, ptts_result* result) {
    return guard([&] {
        if (!d || !d->d || !req || !result) throw Error(PTTS_EINVAL, "dispatcher: nil argument");
        std::string err;
        const int rc = dispatcher_generate(d->d, req, result, &err);
        if (rc != PTTS_OK) throw Error(rc, err.empty() ? "generate: request failed" : err);
    });
}

void ptts_dispatcher_stats(ptts_dispatcher* d, ptts_dispatch_stats* out) {
    if (d && d->d && out) dispatcher_stats(d->d, out);
}

void ptts_dispatcher_close(ptts_dispatcher* d) {
    if (!d) return;
    dispatcher_close(d->d);
    delete d;
}

void ptts_wav_header_streaming(uint8_t out[44]) {   // internal/audio/wav_stream.go:15-41
    static const uint8_t hdr[44] = {'R', 'I', 'F', 'F', 0xFF, 0xFF, 0xFF, 0xFF, 'W', 'A', 'V', 'E', 'f', 'm', 't', ' ', 16, 0, 0, 0, 1, 0, 1, 0,
                                    0xC0, 0x5D, 0, 0 /* 24000 */, 0x80, 0xBB, 0, 0 /* 48000 B/s */, 2, 0, 16, 0, 'd', 'a', 't', 'a', 0xFF, 0xFF, 0xFF, 0xFF};
    memcpy(out, hdr, 44);
}

int ptts_voice_create(ptts_model* h, const float* const* caches, const int64_t* steps, const int64_t* offsets, ptts_voice** out) {
    return guard([&] {
        if (!h || !h->m || !out) throw Error(PTTS_EINVAL, "native: model flow_lm unavailable");
        if (!caches || !steps || !offsets) throw Error(PTTS_EINVAL, "native: voice model state is nil");
        std::lock_guard<std::mutex> lock(h->m->mu);
        h->m->use_device();
        *out = reinterpret_cast<ptts_voice*>(voice_create(*h->m, caches, steps, offsets));
    });
}

void ptts_voice_free(ptts_voice* v) {
    if (!v) return;
    Voice* vv = reinterpret_cast<Voice*>(v);
    (void)hipSetDevice(vv->m->device);
    delete vv;
}

// ---- voice files (voicefile.cpp) ----
int ptts_voice_file_open(const char* path, ptts_voice_file** out) {
    return guard([&] {
        if (!path || !out) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        *out = reinterpret_cast<ptts_voice_file*>(voice_file_from_path(path));
    });
}

int ptts_voice_file_open_bytes(const void* data, size_t len, ptts_voice_file** out) {
    return guard([&] {
        if ((!data && len) || !out) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        *out = reinterpret_cast<ptts_voice_file*>(voice_file_from_bytes(data, len));
    });
}

void ptts_voice_file_close(ptts_voice_file* f) { delete reinterpret_cast<VoiceFile*>(f); }

int32_t ptts_voice_file_kind(const ptts_voice_file* f) { return f ? reinterpret_cast<const VoiceFile*>(f)->kind : PTTS_VOICE_FILE_UNKNOWN; }

int ptts_voice_file_embedding(const ptts_voice_file* f, const float** data, int64_t shape[3]) {
    return guard([&] {
        if (!f || !data || !shape) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        voice_file_embedding(*reinterpret_cast<const VoiceFile*>(f), data, shape);
    });
}

int ptts_voice_file_modules(const ptts_voice_file* f, int32_t* n) {
    return guard([&] {
        if (!f || !n) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        const VoiceFile& v = *reinterpret_cast<const VoiceFile*>(f);
        voice_file_require_state(v);
        *n = (int32_t)v.modules.size();
    });
}

int ptts_voice_file_module(const ptts_voice_file* f, int32_t i, const char** name, ptts_voice_tensor* cache, ptts_voice_tensor* offset) {
    return guard([&] {
        if (!f) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        const VoiceFile& v = *reinterpret_cast<const VoiceFile*>(f);
        voice_file_require_state(v);
        if (i < 0 || (size_t)i >= v.modules.size()) throw Error(PTTS_EINVAL, strfmt("ptts-hip: voice module index %d out of range [0,%zu)", i, v.modules.size()));
        const VoiceFile::Module& m = v.modules[(size_t)i];
        if (name) *name = m.name.c_str();
        auto put = [&](const char* key, ptts_voice_tensor* o) {
            if (!o) return;
            std::memset(o, 0, sizeof *o);
            auto it = m.tensors.find(key);
            if (it == m.tensors.end()) return;
            static const float none = 0.0f;
            o->data = it->second.data.empty() ? &none : it->second.data.data();   // present but empty: data non-NULL, count 0
            o->count = (int64_t)it->second.data.size();
            o->rank = (int32_t)std::min<size_t>(it->second.shape.size(), 8);
            for (int32_t d = 0; d < o->rank; d++) o->shape[d] = it->second.shape[(size_t)d];
        };
        put("cache", cache);
        put("offset", offset);
    });
}

int ptts_voice_file_state(const ptts_voice_file* f, int32_t n_layers, int32_t heads, int32_t head_dim, const float** caches, int64_t* steps, int64_t* offsets) {
    return guard([&] {
        if (!f) throw Error(PTTS_EINVAL, "native: voice model state is nil");
        if (n_layers < 0 || (n_layers > 0 && (!caches || !steps || !offsets))) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        voice_file_state(*reinterpret_cast<const VoiceFile*>(f), n_layers, heads, head_dim, caches, steps, offsets);
    });
}

static void voice_upload_from_file(ptts_model* h, const VoiceFile& vf, ptts_voice** out) {
    Model& m = *h->m;
    std::vector<const float*> caches((size_t)m.d.n_layers);
    std::vector<int64_t> steps((size_t)m.d.n_layers), offs((size_t)m.d.n_layers);
    voice_file_state(vf, m.d.n_layers, m.d.heads, m.d.hd, caches.data(), steps.data(), offs.data());
    std::lock_guard<std::mutex> lock(m.mu);
    m.use_device();
    *out = reinterpret_cast<ptts_voice*>(voice_create(m, caches.data(), steps.data(), offs.data()));
}

int ptts_voice_open(ptts_model* h, const char* path, ptts_voice** out) {
    return guard([&] {
        if (!h || !h->m || !out) throw Error(PTTS_EINVAL, "native: model flow_lm unavailable");
        if (!path) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        std::unique_ptr<VoiceFile> vf(voice_file_from_path(path));
        voice_upload_from_file(h, *vf, out);
    });
}

int ptts_voice_open_bytes(ptts_model* h, const void* data, size_t len, ptts_voice** out) {
    return guard([&] {
        if (!h || !h->m || !out) throw Error(PTTS_EINVAL, "native: model flow_lm unavailable");
        if (!data && len) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        std::unique_ptr<VoiceFile> vf(voice_file_from_bytes(data, len));
        voice_upload_from_file(h, *vf, out);
    });
}

int ptts_profile_enable(ptts_model* h, int32_t on) {
    return guard([&] {
        if (!h || !h->m) throw Error(PTTS_EINVAL, "native-safetensors runtime unavailable");
        std::lock_guard<std::mutex> lock(h->m->mu);
        h->m->prof.on = on == 1;          // 1: per-launch events (plain launches) + phases; 2: phases only
        h->m->prof.phases_on = on != 0;
        h->m->prof.used = 0; h->m->prof.bytes = 0; h->m->prof.wbytes = 0; h->m->prof.launches = 0; h->m->prof.phases = false;
    });
}

int ptts_profile_read(ptts_model* h, ptts_profile* out) {
    return guard([&] {
        if (!h || !h->m || !out) throw Error(PTTS_EINVAL, "native-safetensors runtime unavailable");
        Model& m = *h->m;
        std::lock_guard<std::mutex> lock(m.mu);
        m.use_device();
        PTTS_HIP(hipStreamSynchronize(m.stream));
        std::memset(out, 0, sizeof *out);
        double ms = 0;
        for (size_t i = 0; i + 1 < m.prof.used; i += 2) {
            float t = 0;
            PTTS_HIP(hipEventElapsedTime(&t, m.prof.ev[i], m.prof.ev[i + 1]));
            ms += t;
        }
        out->launches = m.prof.launches;
        out->total_ms = ms;
        out->algorithmic_bytes = m.prof.bytes;
        out->weight_bytes = m.prof.wbytes;
        if (m.prof.phases) {
            PTTS_HIP(hipStreamSynchronize(m.stream2));
            float t = 0;
            PTTS_HIP(hipEventElapsedTime(&t, m.prof.phase[0], m.prof.phase[1])); out->prefill_ms = t;
            PTTS_HIP(hipEventElapsedTime(&t, m.prof.phase[1], m.prof.phase[2])); out->ar_loop_ms = t;
            PTTS_HIP(hipEventElapsedTime(&t, m.prof.phase[3], m.prof.phase[4])); out->mimi_ms = t;
        }
        snprintf(out->kernel, sizeof out->kernel, "%s", "k_skinny");
        m.prof.used = 0; m.prof.bytes = 0; m.prof.wbytes = 0; m.prof.launches = 0; m.prof.phases = false;
    });
}

int ptts_text_embeddings(ptts_model* h, const int64_t* ids, int64_t n, float* out) {
    return guard([&] {
        if (!h || !h->m) throw Error(PTTS_EINVAL, "native: model flow_lm unavailable");
        Model& m = *h->m;
        std::lock_guard<std::mutex> lock(m.mu);
        m.use_device();
        for (int64_t i = 0; i < n; i++)
            if (ids[i] < 0 || ids[i] >= m.d.n_bins) throw Error(PTTS_EINVAL, strfmt("native: token id %lld (%lld) out of range [0,%d)", (long long)i, (long long)ids[i], m.d.n_bins));
        if (n == 0) return;
        DevBuf& dids = m.work(6, (size_t)n * sizeof(int64_t));
        DevBuf& rows = m.work(5, (size_t)n * m.d.d_model * sizeof(float));
        PTTS_HIP(hipMemcpyAsync(dids.p, ids, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, m.stream));
        launch_embed_gather(m.at<float>(m.d.embed), dids.as<int64_t>(), (int)n, m.d.d_model, rows.as<float>(), m.stream);
        PTTS_HIP(hipMemcpyAsync(out, rows.p, (size_t)n * m.d.d_model * sizeof(float), hipMemcpyDeviceToHost, m.stream));
        PTTS_HIP(hipStreamSynchronize(m.stream));
    });
}

int ptts_batch_new(ptts_model* h, int32_t n_slots, int32_t kv_capacity, ptts_batch** out) {
    return guard([&] {
        if (!h || !h->m || !out) throw Error(PTTS_EINVAL, "native: model flow_lm unavailable");
        std::lock_guard<std::mutex> lock(h->m->mu);
        std::unique_ptr<ptts_batch> b(new ptts_batch());
        b->m = h->m;
        b->b = batch_new(*h->m, n_slots, kv_capacity, 1);
        PTTS_HIP(hipStreamSynchronize(h->m->stream));
        *out = b.release();
    });
}

void ptts_batch_free(ptts_batch* b) {
    if (!b) return;
    if (b->b) {
        (void)hipSetDevice(b->m->device);
        (void)hipStreamSynchronize(b->m->stream);
        delete b->b;
    }
    delete b;
}

int ptts_batch_reset(ptts_batch* b) {
    return guard([&] {
        if (!b || !b->b) throw Error(PTTS_EINVAL, "native: flow_lm state unavailable");
        std::lock_guard<std::mutex> lock(b->m->mu);
        b->m->use_device();
        batch_reset(*b->b);
        PTTS_HIP(hipStreamSynchronize(b->m->stream));
    });
}

int ptts_batch_set_voice_state(ptts_batch* b, int32_t slot, const float* const* caches, const int64_t* steps, const int64_t* offsets) {
    return guard([&] {
        if (!b || !b->b) throw Error(PTTS_EINVAL, "native: flow_lm state unavailable");
        if (!caches || !steps || !offsets) throw Error(PTTS_EINVAL, "native: voice model state is nil");
        std::lock_guard<std::mutex> lock(b->m->mu);
        b->m->use_device();
        batch_set_voice(*b->b, slot, caches, steps, offsets);
    });
}

int ptts_batch_prompt(ptts_batch* b, const float* emb, const int64_t* row_offsets) {
    return guard([&] {
        if (!b || !b->b) throw Error(PTTS_EINVAL, "native: flow_lm state unavailable");
        if (!row_offsets) throw Error(PTTS_EINVAL, "native: prompt text embeddings are nil");
        Model& m = *b->m;
        std::lock_guard<std::mutex> lock(m.mu);
        m.use_device();
        int64_t R = row_offsets[b->b->B];
        if (R > 0 && !emb) throw Error(PTTS_EINVAL, "native: prompt text embeddings are nil");
        DevBuf& rows = m.work(5, (size_t)std::max<int64_t>(R, 1) * m.d.d_model * sizeof(float));
        if (R > 0) {
            PTTS_HIP(hipMemcpyAsync(rows.p, emb, (size_t)R * m.d.d_model * sizeof(float), hipMemcpyHostToDevice, m.stream));
            PTTS_HIP(hipStreamSynchronize(m.stream));
        }
        batch_prompt(*b->b, rows.as<float>(), row_offsets);
        PTTS_HIP(hipStreamSynchronize(m.stream));
    });
}

int ptts_batch_step(ptts_batch* hb, const float* frames_in, int32_t lsd_steps, const float* noise, float* frames_out,
                    float* eos_logits, float* last_hidden) {
    return guard([&] {
        if (!hb || !hb->b) throw Error(PTTS_EINVAL, "native: flow_lm state unavailable");
        if (lsd_steps <= 0) throw Error(PTTS_EINVAL, "native: lsd decode steps must be >0");
        if (!frames_in) throw Error(PTTS_EINVAL, "native: replaceNaNWithVector requires non-nil tensors");
        Model& m = *hb->m;
        Batch& b = *hb->b;
        std::lock_guard<std::mutex> lock(m.mu);
        m.use_device();
        hipStream_t s = m.stream;
        const int B = b.B, ld = m.d.ldim;
        for (int i = 0; i < B; i++)
            if (b.kv_len_host[i] + 1 > b.cap) throw Error(PTTS_EINVAL, "ptts-hip: KV capacity exhausted");
        m.tcomb_for(lsd_steps);
        PTTS_HIP(hipMemcpyAsync(b.in_raw.p, frames_in, (size_t)B * ld * sizeof(float), hipMemcpyHostToDevice, s));
        launch_replace_nan(b.in_raw.as<float>(), m.at<float>(m.d.bos), B, ld, b.in32.as<float>(), s);
        if (noise) PTTS_HIP(hipMemcpyAsync(b.cur.p, noise, (size_t)B * ld * sizeof(float), hipMemcpyHostToDevice, s));
        else PTTS_HIP(hipMemsetAsync(b.cur.p, 0, (size_t)B * ld * sizeof(float), s));
        // (the Euler state as the step found it: if a hand-off inside k_flow_cluster times out, the flow part of THIS step is re-issued from it as launches)
        if (b.fc_ok) PTTS_HIP(hipMemcpyAsync(b.cur2.p, b.cur.p, (size_t)B * ld * sizeof(float), hipMemcpyDeviceToDevice, s));
        const bool clustered = b.fc_ok;
        step_core(b, lsd_steps);
        if (clustered) {
            unsigned fault = 0;
            PTTS_HIP(hipMemcpyAsync(&fault, b.fc_fault(), sizeof fault, hipMemcpyDeviceToHost, s));
            PTTS_HIP(hipStreamSynchronize(s));
            if (fault) {   // transformer state (keys, values, `last`, the EOS logit) is untouched by the cluster: only the frame is redone
                flow_cluster_recover(b);
                PTTS_HIP(hipMemcpyAsync(b.cur.p, b.cur2.p, (size_t)B * ld * sizeof(float), hipMemcpyDeviceToDevice, s));
                step_flow_again(b, lsd_steps);
            }
        }
        for (int i = 0; i < B; i++) b.kv_len_host[i] += 1;
        PTTS_HIP(hipMemcpyAsync(b.st.kv_len, b.kv_len_host.data(), (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, s));
        if (frames_out) PTTS_HIP(hipMemcpyAsync(frames_out, b.cur.p, (size_t)B * ld * sizeof(float), hipMemcpyDeviceToHost, s));
        if (eos_logits) PTTS_HIP(hipMemcpyAsync(eos_logits, b.eos.p, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, s));
        if (last_hidden) PTTS_HIP(hipMemcpyAsync(last_hidden, b.last.p, (size_t)B * m.d.d_model * sizeof(float), hipMemcpyDeviceToHost, s));
        PTTS_HIP(hipStreamSynchronize(s));
    });
}

int ptts_batch_offsets(ptts_batch* b, int64_t* out) {
    return guard([&] {
        if (!b || !b->b || !out) throw Error(PTTS_EINVAL, "native: flow_lm state unavailable");
        for (int i = 0; i < b->b->B; i++) out[i] = b->b->kv_len_host[i];
    });
}

int ptts_batch_read_kv(ptts_batch* hb, int32_t slot, int32_t layer, float* k, float* v) {
    return guard([&] {
        if (!hb || !hb->b) throw Error(PTTS_EINVAL, "native: flow_lm state unavailable");
        Model& m = *hb->m;
        Batch& b = *hb->b;
        const Desc& d = m.d;
        if (slot < 0 || slot >= b.B || layer < 0 || layer >= d.n_layers) throw Error(PTTS_EINVAL, "ptts-hip: slot/layer out of range");
        std::lock_guard<std::mutex> lock(m.mu);
        m.use_device();
        const int n = b.kv_len_host[slot];
        const size_t es = b.kv_elem();
        std::vector<uint8_t> tmp((size_t)n * d.hd * es);
        for (int which = 0; which < 2; which++) {
            const char* base = (const char*)(which ? b.vc(layer) : b.kc(layer));
            float* dst = which ? v : k;
            for (int h = 0; h < d.heads; h++) {
                const char* src = base + (((size_t)slot * d.heads + h) * b.cap) * d.hd * es;
                PTTS_HIP(hipMemcpy(tmp.data(), src, tmp.size(), hipMemcpyDeviceToHost));
                float* o = dst + (size_t)h * n * d.hd;
                if (es == 4) std::memcpy(o, tmp.data(), tmp.size());
                else for (size_t i = 0; i < (size_t)n * d.hd; i++) { uint16_t bb; std::memcpy(&bb, tmp.data() + 2 * i, 2); o[i] = bf16_to_f32(bb); }
            }
        }
    });
}

int ptts_decode_latents(ptts_model* h, const float* latents, int32_t n_utt, int32_t frames, float* pcm, float* mimi_latent) {
    return ptts::capi::decode_stages(h, latents, n_utt, frames, pcm, mimi_latent, nullptr);
}

}  // extern "C"

// LatentToMimi + MimiDecode with the staged observation point of tests (transformer_out; capi_hooks.cpp ptts_decode_stages)
int ptts::capi::decode_stages(ptts_model* h, const float* latents, int32_t n_utt, int32_t frames, float* pcm, float* mimi_latent, float* transformer_out) {
    return guard([&] {
        if (!h || !h->m) throw Error(PTTS_EINVAL, "native: model is not fully initialized");
        if (!latents) throw Error(PTTS_EINVAL, "native: latent tensor is nil");
        if (n_utt <= 0 || frames <= 0) throw Error(PTTS_EINVAL, strfmt("native: latent shape must be positive, got [%d %d 32]", n_utt, frames));
        Model& m = *h->m;
        std::lock_guard<std::mutex> lock(m.mu);
        m.use_device();
        const Desc& d = m.d;
        size_t nl = (size_t)n_utt * frames * d.ldim, np = (size_t)n_utt * frames * d.samples_per_frame, nm = (size_t)n_utt * d.mimi_dim * frames;
        size_t nx = transformer_out ? (size_t)n_utt * frames * d.up_stride * d.mimi_dim : 0;
        DevBuf& io = m.work(8, (nl + np + nm + nx) * sizeof(float));
        float* dl = io.as<float>();
        float* dp = dl + nl;
        float* dm = dp + np;
        float* dx = dm + nm;
        PTTS_HIP(hipMemcpyAsync(dl, latents, nl * sizeof(float), hipMemcpyHostToDevice, m.stream));
        mimi_decode(m, dl, (int64_t)frames * d.ldim, n_utt, frames, dp, mimi_latent ? dm : nullptr, transformer_out ? dx : nullptr);
        if (pcm) PTTS_HIP(hipMemcpyAsync(pcm, dp, np * sizeof(float), hipMemcpyDeviceToHost, m.stream));
        if (mimi_latent) PTTS_HIP(hipMemcpyAsync(mimi_latent, dm, nm * sizeof(float), hipMemcpyDeviceToHost, m.stream));
        if (transformer_out) PTTS_HIP(hipMemcpyAsync(transformer_out, dx, nx * sizeof(float), hipMemcpyDeviceToHost, m.stream));
        PTTS_HIP(hipStreamSynchronize(m.stream));
    });
}

extern "C" {

int ptts_speaker_project(ptts_model* h, const float* latent, int64_t frames, float* out) {
    return guard([&] {
        if (!h || !h->m) throw Error(PTTS_EINVAL, "native-safetensors runtime unavailable");
        Model& m = *h->m;
        const Lin& l = m.d.speaker_proj;
        if (l.w == NONE) throw Error(PTTS_EFORMAT, "load speaker_proj_weight: tensor not found in the model weights");
        if (!latent || !out || frames <= 0) throw Error(PTTS_EINVAL, strfmt("latent shape must be [1,T,%d], got [1 %lld %d]", l.in, (long long)frames, l.in));
        std::lock_guard<std::mutex> lock(m.mu);
        m.use_device();
        const size_t ni = (size_t)frames * l.in, no = (size_t)frames * l.out;
        DevBuf& io = m.work(8, (ni + no) * sizeof(float));
        float* di = io.as<float>();
        float* dout = di + ni;
        PTTS_HIP(hipMemcpyAsync(di, latent, ni * sizeof(float), hipMemcpyHostToDevice, m.stream));
        GemmArgs g;
        g.A = di; g.amap = RowMap{l.in, 0, 0};
        g.W = m.arena + l.w; g.w_bf16 = 0; g.ldw = l.in;
        g.C = dout; g.cmap = RowMap{l.out, 0, 0};
        g.M = (int)frames; g.N = l.out; g.K = l.in;
        launch_gemm(g, m.stream);
        PTTS_HIP(hipMemcpyAsync(out, dout, no * sizeof(float), hipMemcpyDeviceToHost, m.stream));
        PTTS_HIP(hipStreamSynchronize(m.stream));
    });
}

int ptts_noise_rows(ptts_model* h, uint64_t seed, float temperature, int32_t rows, float* out) {
    return guard([&] {
        if (!h || !h->m || !out) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        if (rows <= 0) throw Error(PTTS_EINVAL, "native: invalid gaussian noise shape");
        Model& m = *h->m;
        std::lock_guard<std::mutex> lock(m.mu);
        m.use_device();
        const int ld = m.d.ldim;
        if (ld % 4) throw Error(PTTS_EINVAL, "ptts-hip: the device noise draw needs a latent width that is a multiple of 4");
        NoiseSpec sp{seed, temperature > 0.0f ? std::sqrt(temperature) : 0.0f, rows};
        DevBuf& sb = m.work(12, sizeof sp);
        DevBuf& ob = m.work(8, (size_t)rows * ld * sizeof(float));
        PTTS_HIP(hipMemcpyAsync(sb.p, &sp, sizeof sp, hipMemcpyHostToDevice, m.stream));
        PTTS_HIP(hipStreamSynchronize(m.stream));
        launch_noise_fill(sb.as<NoiseSpec>(), 1, rows, ob.as<float>(), (int64_t)rows * ld, ld, m.stream);
        PTTS_HIP(hipMemcpyAsync(out, ob.p, (size_t)rows * ld * sizeof(float), hipMemcpyDeviceToHost, m.stream));
        PTTS_HIP(hipStreamSynchronize(m.stream));
    });
}

int ptts_flow_direction(ptts_model* h, const float* c, float sv, float tv, const float* x, int32_t n, float* out) {
    return guard([&] {
        if (!h || !h->m) throw Error(PTTS_EINVAL, "native: flow_lm flow net unavailable");
        Model& m = *h->m;
        std::lock_guard<std::mutex> lock(m.mu);
        m.use_device();
        // one Euler step of size 1 starting from zero velocity accumulation: run step-core's flow part through a 1-slot-per-row batch
        const Desc& d = m.d;
        const int C = d.flow_dim, D = d.d_model, NA = d.ada_all.out, B = n;
        hipStream_t s = m.stream;
        DevBuf& wsb = m.work(9, ((size_t)B * (D + d.ldim + d.ldim + 4 * C + NA) + C) * sizeof(float));
        float* dc = wsb.as<float>();
        float* dx = dc + (size_t)B * D;
        float* dout = dx + (size_t)B * d.ldim;
        float* sy = dout + (size_t)B * d.ldim;
        float* fx = sy + (size_t)B * C;
        float* fh = fx + (size_t)B * C;
        float* fh2 = fh + (size_t)B * C;
        float* ada = fh2 + (size_t)B * C;
        float* tc = ada + (size_t)B * NA;
        PTTS_HIP(hipMemcpyAsync(dc, c, (size_t)B * D * sizeof(float), hipMemcpyHostToDevice, s));
        PTTS_HIP(hipMemcpyAsync(dx, x, (size_t)B * d.ldim * sizeof(float), hipMemcpyHostToDevice, s));
        PTTS_HIP(hipStreamSynchronize(s));
        m.compute_tcomb(sv, tv, tc);
        auto mkg = [&](const float* A, int64_t lda, const Lin& l, float* Cc, int64_t ldc) {
            GemmArgs g;
            g.A = A; g.amap = RowMap{lda, 0, 0};
            g.W = m.arena + l.w; g.w_bf16 = l.bf16; g.ldw = l.in; g.bias = m.at<float>(l.b);
            g.wt_i8 = l.wt_i8; g.wscale = m.at<float>(l.wscale);
            g.C = Cc; g.cmap = RowMap{ldc, 0, 0};
            g.M = B; g.N = l.out; g.K = l.in;
            return g;
        };
        GemmArgs gc = mkg(dc, D, d.cond_embed, sy, C);
        gc.epi = EPI_SILU; gc.addvec = tc;
        launch_gemm(gc, s);
        launch_gemm(mkg(sy, C, d.ada_all, ada, NA), s);
        launch_gemm(mkg(dx, d.ldim, d.input_proj, fx, C), s);
        for (int r = 0; r < d.flow_depth; r++) {
            const auto& rb = d.rb[r];
            LnArgs ln;
            ln.x = fx; ln.xmap = RowMap{C, 0, 0}; ln.w = m.at<float>(rb.ln.w); ln.b = m.at<float>(rb.ln.b); ln.eps = rb.ln.eps;
            ln.y = fh; ln.ldy = C; ln.rows = B; ln.d = C;
            ln.shift = ada + (size_t)r * 3 * C; ln.scale = ln.shift + C; ln.ldmod = NA;
            launch_layernorm(ln, s);
            GemmArgs g0 = mkg(fh, C, rb.mlp0, fh2, C);
            g0.epi = EPI_SILU;
            launch_gemm(g0, s);
            GemmArgs g2 = mkg(fh2, C, rb.mlp2, fx, C);
            g2.R = fx; g2.epi = EPI_GATE_RESADD; g2.gate = ada + (size_t)r * 3 * C + 2 * C; g2.ldg = NA;
            launch_gemm(g2, s);
        }
        LnArgs lf;
        lf.x = fx; lf.xmap = RowMap{C, 0, 0}; lf.eps = 1e-6f; lf.y = fh; lf.ldy = C; lf.rows = B; lf.d = C;
        lf.shift = ada + (size_t)d.flow_depth * 3 * C; lf.scale = lf.shift + C; lf.ldmod = NA;
        launch_layernorm(lf, s);
        launch_gemm(mkg(fh, C, d.final_linear, dout, d.ldim), s);
        PTTS_HIP(hipMemcpyAsync(out, dout, (size_t)B * d.ldim * sizeof(float), hipMemcpyDeviceToHost, s));
        PTTS_HIP(hipStreamSynchronize(s));
    });
}

// ---- kernel-level entry points ----

int ptts_op_linear(const float* x, const float* w, const float* bias, int64_t rows, int64_t in, int64_t out, float* y) {
    return guard([&] {
        require_device();
        if (!x || !w || !y) throw Error(PTTS_EINVAL, "tensor: linear requires non-nil x and weight");
        Tmp dx((size_t)rows * in * 4), dw((size_t)out * in * 4), db((size_t)out * 4), dy((size_t)rows * out * 4);
        up(dx.p, x, (size_t)rows * in * 4); up(dw.p, w, (size_t)out * in * 4);
        if (bias) up(db.p, bias, (size_t)out * 4);
        GemmArgs g;
        g.A = dx.as<float>(); g.amap = RowMap{in, 0, 0};
        g.W = dw.p; g.ldw = in; g.bias = bias ? db.as<float>() : nullptr;
        g.C = dy.as<float>(); g.cmap = RowMap{out, 0, 0};
        g.M = (int)rows; g.N = (int)out; g.K = (int)in;
        launch_gemm(g, nullptr);
        PTTS_HIP(hipDeviceSynchronize());
        down(y, dy.p, (size_t)rows * out * 4);
    });
}

int ptts_op_pcm16(const float* samples, int64_t n, int16_t* out) {
    return guard([&] {
        require_device();
        if (n < 0 || (n > 0 && (!samples || !out))) throw Error(PTTS_EINVAL, "audio: pcm16 requires non-nil buffers");
        if (n == 0) return;
        Tmp dx((size_t)n * 4), dy((size_t)n * 2);
        up(dx.p, samples, (size_t)n * 4);
        launch_pcm16(dx.as<float>(), dy.as<int16_t>(), n, nullptr);
        PTTS_HIP(hipDeviceSynchronize());
        down(out, dy.p, (size_t)n * 2);
    });
}

int ptts_op_layernorm(const float* x, const float* w, const float* b, float eps, int64_t rows, int64_t d, float* y) {
    return guard([&] {
        require_device();
        if (!x || !y) throw Error(PTTS_EINVAL, "tensor: layernorm input is nil");
        if (eps <= 0) throw Error(PTTS_EINVAL, "tensor: layernorm eps must be > 0");
        if (d <= 0) throw Error(PTTS_EINVAL, "tensor: layernorm last dimension must be > 0");
        Tmp dx((size_t)rows * d * 4), dw((size_t)d * 4), db((size_t)d * 4), dy((size_t)rows * d * 4);
        up(dx.p, x, (size_t)rows * d * 4);
        if (w) up(dw.p, w, (size_t)d * 4);
        if (b) up(db.p, b, (size_t)d * 4);
        LnArgs a;
        a.x = dx.as<float>(); a.xmap = RowMap{d, 0, 0};
        a.w = w ? dw.as<float>() : nullptr; a.b = b ? db.as<float>() : nullptr; a.eps = eps;
        a.y = dy.as<float>(); a.ldy = d; a.rows = (int)rows; a.d = (int)d;
        launch_layernorm(a, nullptr);
        PTTS_HIP(hipDeviceSynchronize());
        down(y, dy.p, (size_t)rows * d * 4);
    });
}

int ptts_op_rope(float* x, const float* cos_t, const float* sin_t, int64_t table_rows, int64_t prefix, int64_t seq, int64_t dim, int64_t pos) {
    return guard([&] {
        require_device();
        if (!x || !cos_t || !sin_t) throw Error(PTTS_EINVAL, "ops: rope requires non-nil x/cos/sin");
        if (pos < 0) throw Error(PTTS_EINVAL, "ops: rope position must be >= 0");
        if (dim % 2) throw Error(PTTS_EINVAL, strfmt("ops: rope last dimension must be even, got %lld", (long long)dim));
        if (table_rows < pos + seq) throw Error(PTTS_EINVAL, strfmt("ops: rope cos/sin sequence length too small for pos=%lld seq=%lld", (long long)pos, (long long)seq));
        size_t n = (size_t)prefix * seq * dim, tn = (size_t)table_rows * (dim / 2);
        Tmp dx(n * 4), dc(tn * 4), ds(tn * 4);
        up(dx.p, x, n * 4); up(dc.p, cos_t, tn * 4); up(ds.p, sin_t, tn * 4);
        launch_rope_rows(dx.as<float>(), RowMap{dim, 0, 0}, 0, 1, (int)dim, nullptr, (int)pos, (int)seq, (int)(prefix * seq), dc.as<float>(), ds.as<float>(), nullptr);
        PTTS_HIP(hipDeviceSynchronize());
        down(x, dx.p, n * 4);
    });
}

int ptts_op_attention_positions(const float* q, const float* k, const float* v, int64_t b, int64_t h, int64_t tq, int64_t tk,
                                int64_t d, const int64_t* posq, const int64_t* posk, int64_t context, float* out) {
    return guard([&] {
        require_device();
        if (!q || !k || !v || !out) throw Error(PTTS_EINVAL, "ops: attention with positions requires non-nil q/k/v");
        if (b <= 0 || h <= 0 || tq <= 0 || tk <= 0 || d <= 0) throw Error(PTTS_EINVAL, "ops: attention expects positive dims");
        if (d > 64) throw Error(PTTS_EINVAL, "ptts-hip: attention kernels are built for head_dim <= 64");
        // cache-slot semantics of the reference's callers (flow_transformer.go:391-420): slot j holds position j or is invalid
        for (int64_t j = 0; j < tk; j++)
            if (posk[j] != -1 && posk[j] != j) throw Error(PTTS_EINVAL, "ptts-hip: posK must be the slot index or -1");
        for (int64_t i = 0; i < tq; i++) {
            int64_t lo = context >= 0 ? std::max<int64_t>(0, posq[i] - context + 1) : 0;
            for (int64_t j = lo; j <= std::min<int64_t>(posq[i], tk - 1); j++)
                if (posk[j] < 0) throw Error(PTTS_EINVAL, "ptts-hip: an invalid key inside a query window is not representable");
            if (posq[i] >= tk) throw Error(PTTS_EINVAL, "ptts-hip: query position beyond the key slots");
        }
        const int HD = 64;
        auto pad = [&](const float* src, int64_t rows) {  // zero-pad head_dim to 64 (dot products are unchanged)
            std::vector<float> o((size_t)rows * HD, 0.0f);
            for (int64_t r = 0; r < rows; r++) for (int64_t e = 0; e < d; e++) { float val = src[r * d + e]; o[(size_t)r * HD + e] = std::isnan(val) ? 0.0f : val; }
            return o;
        };
        // NaN in never-visible slots is allowed by the reference (masked before the dot product); padding drops it too
        std::vector<float> qp = pad(q, b * h * tq), kp = pad(k, b * h * tk), vp = pad(v, b * h * tk);
        const float fix = std::sqrt(64.0f / (float)d);  // kernel scales by 1/sqrt(64); the reference by 1/sqrt(d)
        for (auto& e : qp) e *= fix;
        std::vector<int32_t> pq((size_t)(b * h * tq)), sg((size_t)(b * h * tq));
        Tmp dq(qp.size() * 4), dk(kp.size() * 4), dv(vp.size() * 4), dout((size_t)b * h * tq * HD * 4), dpos(pq.size() * 4), dseg(sg.size() * 4);
        for (int64_t bh = 0; bh < b * h; bh++) for (int64_t i = 0; i < tq; i++) { pq[(size_t)(bh * tq + i)] = (int32_t)posq[i]; sg[(size_t)(bh * tq + i)] = (int32_t)bh; }
        up(dq.p, qp.data(), qp.size() * 4); up(dk.p, kp.data(), kp.size() * 4); up(dv.p, vp.data(), vp.size() * 4);
        up(dpos.p, pq.data(), pq.size() * 4); up(dseg.p, sg.data(), sg.size() * 4);
        AttnArgs a;
        a.q = dq.as<float>(); a.q_ld = HD; a.q_col0 = 0;
        a.k = dk.p; a.v = dv.p;
        a.k_seg_stride = tk * HD; a.k_head_stride = 0; a.k_row_stride = HD;
        a.row_seg = dseg.as<int32_t>(); a.row_pos = dpos.as<int32_t>();
        a.context = (int)context;
        a.out = dout.as<float>(); a.out_ld = HD;
        a.rows = (int)(b * h * tq); a.heads = 1; a.max_keys = (int)tk;
        // Self-attention over a whole sequence with a context window -- positions 0..T-1 on both sides, every slot valid: the
        // call shape of mimiTransformerLayer.selfAttention (mimi.go:365-441).  Implicit positions and segments, so that the
        // launch takes the Mimi decoder's own kernel (k_attn_window) and the reference's context-window tests exercise it.
        bool plain_window = context > 0 && tq == tk;
        for (int64_t i = 0; i < tq && plain_window; i++) plain_window = posq[i] == i && posk[i] == i;
        if (plain_window) {
            a.row_seg = nullptr; a.row_pos = nullptr;
            a.rows_per_seg = (int)tq;
            a.max_keys = (int)std::min<int64_t>(tk, context);
        }
        launch_attention(a, nullptr);
        PTTS_HIP(hipDeviceSynchronize());
        std::vector<float> op((size_t)b * h * tq * HD);
        down(op.data(), dout.p, op.size() * 4);
        for (int64_t r = 0; r < b * h * tq; r++) for (int64_t e = 0; e < d; e++) out[r * d + e] = op[(size_t)r * HD + e];
    });
}

int ptts_op_conv1d_leftpad(const float* x, const float* w, const float* bias, int64_t b, int64_t cin, int64_t len, int64_t cout,
                           int64_t k, float* y) {
    return guard([&] {
        require_device();
        if (!x || !w || !y) throw Error(PTTS_EINVAL, "ops: conv1d requires non-nil input/kernel");
        if (b <= 0 || cin <= 0 || len <= 0 || cout <= 0 || k <= 0) throw Error(PTTS_EINVAL, "ops: conv1d expects positive dims");
        const int64_t P = k - 1;
        std::vector<float> wg((size_t)cout * cin * k);  // [oc][kx*Cin + ic]  (same repack as model.cpp conv_as_gemm)
        for (int64_t o = 0; o < cout; o++) for (int64_t c = 0; c < cin; c++) for (int64_t xk = 0; xk < k; xk++) wg[(size_t)(o * cin * k + xk * cin + c)] = w[(o * cin + c) * k + xk];
        Tmp dx((size_t)b * cin * len * 4), dxc((size_t)b * (P + len) * cin * 4), dw(wg.size() * 4), db((size_t)cout * 4),
            dyc((size_t)b * len * cout * 4), dy((size_t)b * cout * len * 4);
        up(dx.p, x, (size_t)b * cin * len * 4); up(dw.p, wg.data(), wg.size() * 4);
        if (bias) up(db.p, bias, (size_t)cout * 4);
        PTTS_HIP(hipMemset(dxc.p, 0, (size_t)b * (P + len) * cin * 4));
        launch_bct_to_btc(dx.as<float>(), (int)b, (int)cin, (int)len, dxc.as<float>(), (int)P, nullptr);
        GemmArgs g;
        g.A = dxc.as<float>(); g.amap = RowMap{cin, len, (P + len) * cin};
        g.W = dw.p; g.ldw = cin * k; g.bias = bias ? db.as<float>() : nullptr;
        g.C = dyc.as<float>(); g.cmap = RowMap{cout, 0, 0};
        g.M = (int)(b * len); g.N = (int)cout; g.K = (int)(cin * k);
        launch_gemm(g, nullptr);
        launch_btc_to_bct(dyc.as<float>(), 0, (int)b, (int)cout, (int)len, dy.as<float>(), nullptr);
        PTTS_HIP(hipDeviceSynchronize());
        down(y, dy.p, (size_t)b * cout * len * 4);
    });
}

int ptts_op_convtr1d_righttrim(const float* x, const float* w, const float* bias, int64_t b, int64_t cin, int64_t len,
                               int64_t opg, int64_t k, int64_t stride, int64_t groups, float* y) {
    return guard([&] {
        require_device();
        if (!x || !w || !y) throw Error(PTTS_EINVAL, "ops: convtranspose1d requires non-nil input/kernel");
        if (stride <= 0 || groups <= 0) throw Error(PTTS_EINVAL, "ops: convtranspose1d stride/dilation/groups must be > 0");
        if (k != 2 * stride) throw Error(PTTS_EINVAL, "ptts-hip: transposed convolutions are built for kernel = 2*stride");
        const int64_t lout = len * stride;
        if (groups == 1) {
            const int64_t cout = opg;
            std::vector<float> wg((size_t)stride * cout * 2 * cin), bg((size_t)stride * cout, 0.0f);
            for (int64_t r = 0; r < stride; r++) for (int64_t o = 0; o < cout; o++) {
                for (int64_t c = 0; c < cin; c++) {
                    wg[(size_t)((r * cout + o) * 2 * cin + c)] = w[(c * cout + o) * k + r + stride];
                    wg[(size_t)((r * cout + o) * 2 * cin + cin + c)] = w[(c * cout + o) * k + r];
                }
                if (bias) bg[(size_t)(r * cout + o)] = bias[o];
            }
            Tmp dx((size_t)b * cin * len * 4), dxc((size_t)b * (1 + len) * cin * 4), dw(wg.size() * 4), db(bg.size() * 4),
                dyc((size_t)b * lout * cout * 4), dy((size_t)b * cout * lout * 4);
            up(dx.p, x, (size_t)b * cin * len * 4); up(dw.p, wg.data(), wg.size() * 4); up(db.p, bg.data(), bg.size() * 4);
            PTTS_HIP(hipMemset(dxc.p, 0, (size_t)b * (1 + len) * cin * 4));
            launch_bct_to_btc(dx.as<float>(), (int)b, (int)cin, (int)len, dxc.as<float>(), 1, nullptr);
            GemmArgs g;
            g.A = dxc.as<float>(); g.amap = RowMap{cin, len, (1 + len) * cin};
            g.W = dw.p; g.ldw = 2 * cin; g.bias = db.as<float>();
            g.C = dyc.as<float>(); g.cmap = RowMap{stride * cout, 0, 0};
            g.M = (int)(b * len); g.N = (int)(stride * cout); g.K = (int)(2 * cin);
            launch_gemm(g, nullptr);
            launch_btc_to_bct(dyc.as<float>(), 0, (int)b, (int)cout, (int)lout, dy.as<float>(), nullptr);
            PTTS_HIP(hipDeviceSynchronize());
            down(y, dy.p, (size_t)b * cout * lout * 4);
            return;
        }
        if (groups == cin && opg == 1) {
            std::vector<float> w0((size_t)stride * cin), w1((size_t)stride * cin);
            for (int64_t r = 0; r < stride; r++) for (int64_t c = 0; c < cin; c++) { w0[(size_t)(r * cin + c)] = w[c * k + r + stride]; w1[(size_t)(r * cin + c)] = w[c * k + r]; }
            Tmp dx((size_t)b * cin * len * 4), dxc((size_t)b * (1 + len) * cin * 4), d0(w0.size() * 4), d1(w1.size() * 4), db((size_t)cin * 4),
                dyc((size_t)b * lout * cin * 4), dy((size_t)b * cin * lout * 4);
            up(dx.p, x, (size_t)b * cin * len * 4); up(d0.p, w0.data(), w0.size() * 4); up(d1.p, w1.data(), w1.size() * 4);
            if (bias) up(db.p, bias, (size_t)cin * 4);
            PTTS_HIP(hipMemset(dxc.p, 0, (size_t)b * (1 + len) * cin * 4));
            launch_bct_to_btc(dx.as<float>(), (int)b, (int)cin, (int)len, dxc.as<float>(), 1, nullptr);
            launch_upsample_depthwise(dxc.as<float>(), d0.as<float>(), d1.as<float>(), bias ? db.as<float>() : nullptr, (int)b, (int)len, 0, (int)len,
                                      (int)cin, (int)stride, dyc.as<float>(), 0, nullptr);
            launch_btc_to_bct(dyc.as<float>(), 0, (int)b, (int)cin, (int)lout, dy.as<float>(), nullptr);
            PTTS_HIP(hipDeviceSynchronize());
            down(y, dy.p, (size_t)b * cin * lout * 4);
            return;
        }
        throw Error(PTTS_EINVAL, "ptts-hip: transposed convolution supports groups == 1 or depthwise (groups == in_channels)");
    });
}

}  // extern "C"
