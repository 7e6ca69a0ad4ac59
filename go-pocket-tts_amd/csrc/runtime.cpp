// runtime.cpp -- engine: model open, FlowLM batch state, prefill, AR step, Mimi decode, GenerateAudio.
// Reference call path being replaced: internal/tts/runtime_native_safetensors.go:52-238 ->
// internal/native/{model,flow_lm,flow_transformer,flow_net,mimi}.go.
#include "runtime.h"

#include <chrono>

#include <algorithm>
#include <cmath>

namespace ptts {

static RowMap flat(int64_t ld) { return RowMap{ld, 0, 0}; }
static RowMap seg(int64_t ld, int64_t rows_per_batch, int64_t batch_stride) { return RowMap{ld, rows_per_batch, batch_stride}; }

static GemmArgs mk(const Model& m, const float* A, RowMap am, const Lin& l, float* C, RowMap cm, int M) {
    GemmArgs g;
    g.A = A; g.amap = am;
    g.W = m.arena + l.w; g.w_bf16 = l.bf16; g.ldw = l.in;
    g.Wt = l.wt == NONE ? nullptr : m.arena + l.wt;
    g.wt_i8 = l.wt_i8; g.wscale = m.at<float>(l.wscale);
    g.bias = m.at<float>(l.b);
    g.C = C; g.cmap = cm;
    g.M = M; g.N = l.out; g.K = l.in;
    return g;
}
static LnArgs mkln(const Model& m, const float* x, RowMap xm, const Norm& n, float* y, int64_t ldy, int rows) {
    LnArgs a;
    a.x = x; a.xmap = xm;
    a.w = m.at<float>(n.w); a.b = m.at<float>(n.b);
    a.eps = n.eps;
    a.y = y; a.ldy = ldy;
    a.rows = rows; a.d = n.d;
    return a;
}

// Small host->device uploads (slot tables, token ids, per-utterance limits): a copy from pageable memory has to be waited for
// (~20 us each, a dozen per call).  Inside generate_chunk they are staged through a page-locked arena instead and queued
// without a wait; the arena is rewound when the call starts (the previous call ended with the stream drained).
static thread_local UploadArena* tl_upload = nullptr;
UploadScope::UploadScope(UploadArena& a, hipStream_t s) {   // uploads of this thread go through the arena until the scope ends
    (void)hipStreamSynchronize(s);   // normally idle already; after a failed call it may still be reading the arena
    if (!a.base && hipHostMalloc((void**)&a.base, (size_t)1 << 20, hipHostMallocDefault) == hipSuccess) a.cap = (size_t)1 << 20;
    else if (!a.base) (void)hipGetLastError();
    a.off = 0;
    tl_upload = &a;
}
UploadScope::UploadScope(UploadArena& a, hipEvent_t last_use) {   // an arena of its own, reused once the copies queued from it last time are done
    if (last_use) (void)hipEventSynchronize(last_use);
    if (!a.base && hipHostMalloc((void**)&a.base, (size_t)1 << 20, hipHostMallocDefault) == hipSuccess) a.cap = (size_t)1 << 20;
    else if (!a.base) (void)hipGetLastError();
    a.off = 0;
    tl_upload = &a;
}
UploadScope::~UploadScope() { tl_upload = nullptr; }
void h2d(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    UploadArena* a = tl_upload;
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (a && a->base && a->off + need <= a->cap) {
        char* stage = a->base + a->off;
        a->off += need;
        std::memcpy(stage, src, bytes);
        PTTS_HIP(hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, s));
        return;
    }
    PTTS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    PTTS_HIP(hipStreamSynchronize(s));
}
void d2h(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    PTTS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    PTTS_HIP(hipStreamSynchronize(s));
}

// ------------------------------------------------------------------------------------------------
// Model
// ------------------------------------------------------------------------------------------------
Model::~Model() {
    if (upload.base) (void)hipHostFree(upload.base);
    for (hipEvent_t e : prof.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : prof.phase) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    if (stream2) (void)hipStreamDestroy(stream2);

    cached_batch.reset();
    tcomb.clear();
    ws.clear();
    if (own_arena && arena) (void)hipFree(arena);
    if (stream) (void)hipStreamDestroy(stream);
}

Model* model_open(Plan* plan, void* device_arena, int fill) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw Error(PTTS_ENODEVICE, "ptts-hip: no HIP device available (this library has no CPU fallback)");
    if (plan->opts.device < 0 || plan->opts.device >= ndev)
        throw Error(PTTS_ENODEVICE, strfmt("ptts-hip: device %d out of range (%d visible)", plan->opts.device, ndev));
    std::unique_ptr<Model> m(new Model());
    m->d = plan->desc;
    m->opts = plan->opts;
    m->device = plan->opts.device;
    m->noise_state = (uint64_t)std::chrono::system_clock::now().time_since_epoch().count() ^ (uint64_t)(uintptr_t)m.get();
    m->use_device();
    {
        int lo = 0, hi = 0;   // numerically lower = higher priority
        PTTS_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        PTTS_HIP(hipStreamCreateWithPriority(&m->stream, hipStreamNonBlocking, hi));
        PTTS_HIP(hipStreamCreateWithPriority(&m->stream2, hipStreamNonBlocking, lo));
    }
    if (device_arena) {
        m->arena = reinterpret_cast<uint8_t*>(device_arena);
    } else {
        void* p = nullptr;
        PTTS_HIP(hipMalloc(&p, m->d.total_bytes));
        m->arena = reinterpret_cast<uint8_t*>(p);
        m->own_arena = true;
        fill = 1;
    }
    if (fill) {
        std::vector<uint8_t> host(m->d.total_bytes, 0);
        plan_fill(*plan, host.data());
        PTTS_HIP(hipMemcpy(m->arena, host.data(), host.size(), hipMemcpyHostToDevice));
    }
    return m.release();
}

// A second engine over the weights of `base`: its own streams, KV caches and workspaces, the SAME arena (read-only after load).
// Two engines on one GPU let one batch's Mimi decode (throughput work) run beside the next batch's prefill and AR loop
// (latency-bound, most of the chip idle): tools/serve_bench.py, 256 clients: 9.1 k -> 12.2 k x real time.
Model* model_share(Model& base) {
    std::unique_ptr<Model> m(new Model());
    m->d = base.d;
    m->opts = base.opts;
    m->device = base.device;
    m->noise_state = (uint64_t)std::chrono::system_clock::now().time_since_epoch().count() ^ (uint64_t)(uintptr_t)m.get();
    m->use_device();
    int lo = 0, hi = 0;
    PTTS_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    PTTS_HIP(hipStreamCreateWithPriority(&m->stream, hipStreamNonBlocking, hi));
    PTTS_HIP(hipStreamCreateWithPriority(&m->stream2, hipStreamNonBlocking, lo));
    m->arena = base.arena;
    m->own_arena = false;
    return m.release();
}

// The model again on ANOTHER GPU of the same process: its own arena, filled from base's over the GPUs' direct link (hipMemcpyPeer: xGMI on an MI355X node) --
// no second file read, no host staging, no collective library.  This is the shape the reference's server has: ONE process holding N workers
// (internal/server/server.go:119-143,398-421; cmd/pockettts/serve.go:15-52); there the workers share one CPU model, here every GPU gets the bytes once at
// start-up and a dispatcher over the N models deals the requests (dispatcher.cpp: one worker thread per model, each on its own device).  `device` may be
// base's own (a second private copy on the same GPU: what a one-GPU box can test).
Model* model_replicate(Model& base, int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw Error(PTTS_ENODEVICE, "ptts-hip: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) throw Error(PTTS_ENODEVICE, strfmt("ptts-hip: device %d out of range (%d visible)", device, ndev));
    std::unique_ptr<Model> m(new Model());
    m->d = base.d;
    m->opts = base.opts;
    m->opts.device = device;
    m->device = device;
    m->noise_state = (uint64_t)std::chrono::system_clock::now().time_since_epoch().count() ^ (uint64_t)(uintptr_t)m.get();
    {   // base's arena is complete and idle-readable: nothing of base may be mid-upload (model_open returns after its copy)
        std::lock_guard<std::mutex> lock(base.mu);
        base.use_device();
        PTTS_HIP(hipStreamSynchronize(base.stream));
    }
    m->use_device();
    int lo = 0, hi = 0;
    PTTS_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    PTTS_HIP(hipStreamCreateWithPriority(&m->stream, hipStreamNonBlocking, hi));
    PTTS_HIP(hipStreamCreateWithPriority(&m->stream2, hipStreamNonBlocking, lo));
    void* p = nullptr;
    PTTS_HIP(hipMalloc(&p, m->d.total_bytes));
    m->arena = reinterpret_cast<uint8_t*>(p);
    m->own_arena = true;
    if (device != base.device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device, base.device) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(base.device, 0);   // (speed only: hipMemcpyPeer stages through the host without it)
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
    }
    PTTS_HIP(hipMemcpyPeer(m->arena, device, base.arena, base.device, m->d.total_bytes));
    PTTS_HIP(hipDeviceSynchronize());
    return m.release();
}

// timestep embedder (flow_net.go:42-83) for one (s, t) pair, then 0.5*(e_s + e_t) (flow_net.go:320-335)
void Model::compute_tcomb(float sv, float tv, float* dst) {
    const int C = d.flow_dim, nf = d.nfreq;
    DevBuf& wsb = work(0, (size_t)(2 * nf + 3 * C) * sizeof(float));
    float* feat = wsb.as<float>();
    float* h = feat + 2 * nf;
    float* e[2] = {h + C, h + 2 * C};
    const float tvals[2] = {sv, tv};
    for (int i = 0; i < 2; i++) {
        const auto& te = d.te[i];
        if (te.l1.in != 2 * nf) throw Error(PTTS_EFORMAT, "native: timestep embedder width mismatch");
        launch_timestep_features(tvals[i], at<float>(te.freqs), nf, feat, stream);
        GemmArgs g1 = mk(*this, feat, flat(2 * nf), te.l1, h, flat(C), 1);
        g1.epi = EPI_SILU;
        launch_gemm(g1, stream);
        GemmArgs g2 = mk(*this, h, flat(C), te.l2, e[i], flat(C), 1);
        launch_gemm(g2, stream);
        launch_rmsnorm_alpha(e[i], at<float>(te.alpha), 1e-5f, 1, C, stream);
    }
    launch_avg2(e[0], e[1], dst, C, stream);
}

const float* Model::tcomb_for(int n) {
    auto it = tcomb.find(n);
    if (it != tcomb.end()) return it->second->as<float>();
    std::unique_ptr<DevBuf> buf(new DevBuf());
    buf->ensure((size_t)n * d.flow_dim * sizeof(float));
    for (int i = 0; i < n; i++)  // flow_lm.go:325-329: s = i/n, t = (i+1)/n in float32
        compute_tcomb((float)i / (float)n, (float)(i + 1) / (float)n, buf->as<float>() + (size_t)i * d.flow_dim);
    PTTS_HIP(hipStreamSynchronize(stream));
    const float* p = buf->as<float>();
    tcomb[n] = std::move(buf);
    return p;
}

// ------------------------------------------------------------------------------------------------
// Batch
// ------------------------------------------------------------------------------------------------
Batch::~Batch() {
    if (fc_stamps.p) {   // (PTTS_FC_STAMPS=<file>: the timestamps of the batch's LAST k_flow_cluster launch, one line per workgroup)
        constexpr int kWg = 8 * kFlowClusterMaxTiles;
        std::vector<unsigned long long> h((size_t)kWg * 64);
        if (const char* path = getenv("PTTS_FC_STAMPS"))
            if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(h.data(), fc_stamps.p, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
                if (FILE* f = fopen(path, "a")) {
                    for (int blk = 0; blk < kWg; blk++) {
                        if (!h[(size_t)blk * 64]) continue;
                        fprintf(f, "wg %d:", blk);
                        for (int i = 0; i < 64; i++) fprintf(f, " %llu", h[(size_t)blk * 64 + i]);
                        fprintf(f, "\n");
                    }
                    fprintf(f, "\n");
                    fclose(f);
                }
    }
    for (auto& set : graphs) for (auto& row : set) for (hipGraphExec_t g : row) if (g) (void)hipGraphExecDestroy(g);
    if (n_active_pinned) (void)hipHostFree(n_active_pinned);
    if (rows_pinned) (void)hipHostFree(rows_pinned);
}

static bool open_linears(Batch& b, StepOpenLinears& lin);
static bool step_flow(Batch& b, int lsd, bool opened, bool fuse_finish, bool chain);

Batch* batch_new(Model& m, int n_slots, int cap, int max_steps) {
    if (n_slots <= 0) throw Error(PTTS_EINVAL, "ptts-hip: batch needs at least one slot");
    if (n_slots > kStepMaxRows) throw Error(PTTS_EINVAL, strfmt("ptts-hip: a batch takes at most %d slots (the AR step's kernels), asked for %d", kStepMaxRows, n_slots));
    if (cap <= 0 || cap > ROPE_SEQ) throw Error(PTTS_EINVAL, strfmt("ptts-hip: kv capacity %d outside (0, %d] (RoPE table rows, flow_transformer.go:505)", cap, ROPE_SEQ));
    m.use_device();
    const Desc& d = m.d;
    std::unique_ptr<Batch> b(new Batch());
    b->m = &m;
    b->B = n_slots;
    b->cap = cap;
    b->max_steps = std::max(1, max_steps);
    const size_t B = (size_t)n_slots;
    size_t kvbytes = (size_t)d.n_layers * B * d.heads * cap * d.hd * b->kv_elem();
    b->kcache.ensure(kvbytes);
    b->vcache.ensure(kvbytes);
    // the step attention re-reads a valid row for key slots past the end (zero weight): every row must hold finite data
    PTTS_HIP(hipMemsetAsync(b->kcache.p, 0, kvbytes, m.stream));
    PTTS_HIP(hipMemsetAsync(b->vcache.p, 0, kvbytes, m.stream));
    b->state_i32.ensure((9 * B + 1) * sizeof(int32_t));
    b->state_f32.ensure(B * sizeof(float));
    int32_t* s = b->state_i32.as<int32_t>();
    b->st.kv_len = s; b->st.active = s + B; b->st.step = s + 2 * B; b->st.countdown = s + 3 * B;
    b->st.n_frames = s + 4 * B; b->st.eos_step = s + 5 * B; b->st.max_steps = s + 6 * B;
    b->st.frames_after_eos = s + 7 * B; b->st.broke = s + 8 * B; b->st.n_active = s + 9 * B;
    b->st.eos_threshold = b->state_f32.as<float>();
    b->kv_len_host.assign(B, 0);
    b->pre_k.ensure((size_t)B * sizeof(void*)); b->pre_v.ensure((size_t)B * sizeof(void*)); b->pre_len.ensure((size_t)B * sizeof(int32_t));
    b->pre_k_host.assign(B, nullptr); b->pre_v_host.assign(B, nullptr); b->pre_len_host.assign(B, 0);
    const size_t f = sizeof(float);
    const int NA = d.ada_all.out;
    b->in_raw.ensure(B * d.ldim * f); b->in32.ensure(B * d.ldim * f);
    b->x.ensure(B * d.d_model * f); b->xn.ensure(B * std::max(d.d_model, d.flow_dim) * f);
    b->qkv.ensure(B * 3 * d.d_model * f); b->attn.ensure(B * d.d_model * f);
    b->ff.ensure(B * d.ffn * f); b->last.ensure(B * d.d_model * f); b->eos.ensure(B * f);
    b->sy.ensure(B * d.flow_dim * f); b->ada.ensure(B * NA * f);
    b->fx.ensure(B * d.flow_dim * f); b->fh.ensure(B * d.flow_dim * f); b->fh2.ensure(B * d.flow_dim * f);
    b->cur.ensure(B * d.ldim * f);
    b->partial.ensure((size_t)16 * B * std::max(d.d_model, d.flow_dim) * f);
    b->latents.ensure(B * b->max_steps * d.ldim * f);
    {
        StepFinish sf{b->st, b->eos.as<float>(), b->latents.as<float>(), (int64_t)b->max_steps * d.ldim, (int32_t)d.ldim, StepChain{}};
        StepOpenLinears lin;
        b->chain_ok = open_linears(*b, lin) && step_open_mfma_ok(lin, d.ldim) && ((int64_t)b->max_steps * d.ldim) % 2 == 0 &&
                      (lin.w_bf16 != 0) == (d.final_linear.bf16 != 0 || d.final_linear.wt_i8 != 0);   // the chained form takes the 32-deep linears' type from the step kernel's weight type
        // two copies: a step that reads fx / cur (par 0) opens the next one in fx2 / cur2, and the other way round
        b->fx2.ensure(B * d.flow_dim * f); b->cur2.ensure(B * d.ldim * f);
        StepFinish sf2[2] = {sf, sf};
        sf2[0].ch = StepChain{lin.w_in, lin.b_in, lin.x, lin.d_in, lin.w_pj, lin.b_pj, b->fx2.as<float>(), lin.d_pj, m.at<float>(d.bos), b->cur2.as<float>(), lin.w_bf16, b->chain_ok ? 1 : 0};
        sf2[1].ch = StepChain{lin.w_in, lin.b_in, lin.x, lin.d_in, lin.w_pj, lin.b_pj, b->fx.as<float>(), lin.d_pj, m.at<float>(d.bos), b->cur.as<float>(), lin.w_bf16, b->chain_ok ? 1 : 0};
        b->fin_dev.ensure(sizeof sf2);
        h2d(b->fin_dev.p, sf2, sizeof sf2, m.stream);
    }
    {   // flow_cluster.hip: bf16 step copies of every residual block's linears, the width it is built for
        const char* sw = getenv("PTTS_FLOW_CLUSTER");   // A/B switch, read per batch: 0 = the 2 x depth launches (tests compare the two forms bit for bit)
        const bool off = sw && sw[0] == '0';
        bool ok = !off && !m.fc_disabled.load() && d.flow_dim == 512 && d.flow_depth > 0 && d.flow_depth <= FC_MAX_DEPTH && n_slots <= kStepMaxRows &&
                  flow_cluster_fits(n_slots, m.device);   // (every workgroup of the grid resident at once: its hand-offs spin on its peers)
        for (int r = 0; ok && r < d.flow_depth; r++) {
            const auto& rb = d.rb[r];
            for (const Lin* l : {&rb.mlp0, &rb.mlp2}) ok = ok && l->in == 512 && l->out == 512 && l->bf16 && !l->wt_i8 && l->wt != NONE && l->b != NONE;
            ok = ok && rb.ln.w != NONE && rb.ln.b != NONE && rb.ln.d == 512;
        }
        b->fc_ok = ok;
        if (ok) {
            b->fc_xbuf.ensure(flow_cluster_xbuf_bytes(n_slots)); b->fc_sync.ensure(kFlowClusterSyncBytes);
            PTTS_HIP(hipMemsetAsync(b->fc_xbuf.p, 0, flow_cluster_xbuf_bytes(n_slots), m.stream));
            PTTS_HIP(hipMemsetAsync(b->fc_sync.p, 0, kFlowClusterSyncBytes, m.stream));
            if (getenv("PTTS_FC_STAMPS")) { b->fc_stamps.ensure((size_t)8 * kFlowClusterMaxTiles * 64 * 8); PTTS_HIP(hipMemsetAsync(b->fc_stamps.p, 0, (size_t)8 * kFlowClusterMaxTiles * 64 * 8, m.stream)); }
        }
    }
    {   // tall.hip: from kTallMinRows rows the layer's in_proj / linear1 / linear2 run as 64 x 64-tile products on bf16 row planes
        const char* sw = getenv("PTTS_TALL");   // A/B switch, read per batch: 0 = k_skinny for every row count
        bool ok = !(sw && sw[0] == '0') && n_slots >= kTallMinRows && (d.d_model == 512 || d.d_model == 1024) && d.ffn % 128 == 0;
        for (int l = 0; ok && l < d.n_layers; l++) {
            const auto& L = d.layers[l];
            for (const Lin* li : {&L.in_proj, &L.l1, &L.l2}) ok = ok && li->bf16 && !li->wt_i8 && li->wt != NONE && li->out % 4 == 0;
            ok = ok && L.in_proj.in == d.d_model && L.l1.in == d.d_model && L.l1.out == d.ffn && L.l2.in == d.ffn && L.l2.out == d.d_model;
            ok = ok && L.n1.w != NONE && L.n1.b != NONE && L.n2.w != NONE && L.n2.b != NONE;
        }
        b->tall_ok = ok;
        if (ok) { b->tp_a.ensure(2 * B * d.d_model * sizeof(uint16_t)); b->tp_f.ensure(2 * B * d.ffn * sizeof(uint16_t)); }
    }
    PTTS_HIP(hipHostMalloc((void**)&b->n_active_pinned, sizeof(int32_t) * (size_t)(2 + 2 * B), hipHostMallocDefault));
    PTTS_HIP(hipHostMalloc((void**)&b->rows_pinned, sizeof(PcmRow) * std::max<size_t>((size_t)B, 1), hipHostMallocDefault));
    batch_reset(*b);
    return b.release();
}

void batch_reset(Batch& b) {
    b.opened = false;
    hipStream_t s = b.m->stream;
    const int B = b.B;
    launch_fill_i32(b.st.kv_len, 0, B, s);
    b.kv_bound = 0;
    launch_fill_i32(b.st.active, 1, B, s);
    launch_fill_i32(b.st.step, 0, B, s);
    launch_fill_i32(b.st.countdown, -1, B, s);
    launch_fill_i32(b.st.n_frames, 0, B, s);
    launch_fill_i32(b.st.eos_step, -1, B, s);
    launch_fill_i32(b.st.max_steps, 0x7fffffff, B, s);
    launch_fill_i32(b.st.frames_after_eos, 0, B, s);
    launch_fill_i32(b.st.broke, 0, B, s);
    launch_fill_i32(b.st.n_active, B, 1, s);
    std::vector<float> thr((size_t)B, INFINITY);
    h2d(b.st.eos_threshold, thr.data(), thr.size() * sizeof(float), s);
    std::fill(b.kv_len_host.begin(), b.kv_len_host.end(), 0);
    std::fill(b.pre_len_host.begin(), b.pre_len_host.end(), 0);
    launch_fill_i32(b.pre_len.as<int32_t>(), 0, B, s);
    b.has_noise = false;
}

static void check_voice(const Desc& d, const float* const* caches, const int64_t* steps, const int64_t* offsets, int64_t cap) {
    for (int l = 0; l < d.n_layers; l++) {  // flow_transformer.go:538-549
        if (!caches[l]) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"transformer.layers.%d.self_attn\" missing cache", l));
        if (offsets[l] < 0) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"transformer.layers.%d.self_attn\" has negative offset %lld", l, (long long)offsets[l]));
        if (offsets[l] > steps[l]) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"transformer.layers.%d.self_attn\" offset %lld exceeds cache length %lld", l, (long long)offsets[l], (long long)steps[l]));
        if (offsets[l] != offsets[0]) throw Error(PTTS_EINVAL, "ptts-hip: per-layer voice offsets differ; one offset per utterance is supported");
        if (offsets[l] > cap) throw Error(PTTS_EINVAL, "ptts-hip: voice state longer than the KV capacity");
    }
}

Voice* voice_create(Model& m, const float* const* caches, const int64_t* steps, const int64_t* offsets) {
    const Desc& d = m.d;
    check_voice(d, caches, steps, offsets, ROPE_SEQ);
    std::unique_ptr<Voice> v(new Voice());
    v->m = &m;
    v->offset = (int)offsets[0];
    const size_t lb = v->layer_bytes();
    v->k.ensure(std::max<size_t>(lb * d.n_layers, 256));
    v->v.ensure(std::max<size_t>(lb * d.n_layers, 256));
    for (int l = 0; l < d.n_layers; l++) {
        size_t n = (size_t)2 * steps[l] * d.heads * d.hd;
        DevBuf& raw = m.work(1, n * sizeof(float));
        h2d(raw.p, caches[l], n * sizeof(float), m.stream);
        launch_voice_scatter(raw.as<float>(), (int)steps[l], d.heads, d.hd, v->offset, 0, (char*)v->k.p + lb * l, (char*)v->v.p + lb * l,
                             m.opts.kv == PTTS_KV_BF16, v->offset, m.stream);
        PTTS_HIP(hipStreamSynchronize(m.stream));
    }
    return v.release();
}

// a voice is device memory in the cache layout: every engine on the same GPU with the same cache geometry can read it
bool voice_usable_by(const Voice& v, const Model& m) {
    return v.m == &m || (v.m && v.m->device == m.device && v.m->opts.kv == m.opts.kv && v.m->d.n_layers == m.d.n_layers && v.m->d.heads == m.d.heads &&
                         v.m->d.hd == m.d.hd);
}

void batch_apply_voice(Batch& b, const Voice& v, const std::vector<int32_t>& slots) {
    Model& m = *b.m;
    hipStream_t io = b.io_stream ? b.io_stream : m.stream;
    const Desc& d = m.d;
    if (!voice_usable_by(v, m)) throw Error(PTTS_EINVAL, "ptts-hip: voice belongs to another model");
    if (v.offset > b.cap) throw Error(PTTS_EINVAL, "ptts-hip: voice state longer than the KV capacity");
    DevBuf& ds = m.work(10, slots.size() * sizeof(int32_t));
    h2d(ds.p, slots.data(), slots.size() * sizeof(int32_t), io);
    const size_t lb = v.layer_bytes();
    for (int l = 0; l < d.n_layers; l++)
        launch_voice_apply((const char*)v.k.p + lb * l, (const char*)v.v.p + lb * l, v.offset, d.heads, d.hd, ds.as<int32_t>(), (int)slots.size(),
                           b.kc(l), b.vc(l), (int)b.kv_elem(), b.cap, io);
    for (int32_t sl : slots) {
        b.kv_len_host[sl] = v.offset;
        b.kv_bound = std::max(b.kv_bound, (int)v.offset);
        b.pre_k_host[sl] = v.k.p; b.pre_v_host[sl] = v.v.p; b.pre_len_host[sl] = v.offset;
    }
    if (b.slot_local) return;   // continuous batch: the admit kernel writes these entries for the slots it fills; the others are live
    h2d(b.st.kv_len, b.kv_len_host.data(), (size_t)b.B * sizeof(int32_t), io);
    h2d(b.pre_k.p, b.pre_k_host.data(), (size_t)b.B * sizeof(void*), io);
    h2d(b.pre_v.p, b.pre_v_host.data(), (size_t)b.B * sizeof(void*), io);
    h2d(b.pre_len.p, b.pre_len_host.data(), (size_t)b.B * sizeof(int32_t), io);
}

void batch_set_voice(Batch& b, int slot, const float* const* caches, const int64_t* steps, const int64_t* offsets) {
    Model& m = *b.m;
    hipStream_t io = b.io_stream ? b.io_stream : m.stream;
    const Desc& d = m.d;
    if (slot < 0 || slot >= b.B) throw Error(PTTS_EINVAL, "ptts-hip: voice state slot out of range");
    check_voice(d, caches, steps, offsets, b.cap);
    for (int l = 0; l < d.n_layers; l++) {
        size_t n = (size_t)2 * steps[l] * d.heads * d.hd;
        DevBuf& raw = m.work(1, n * sizeof(float));
        h2d(raw.p, caches[l], n * sizeof(float), io);
        launch_voice_scatter(raw.as<float>(), (int)steps[l], d.heads, d.hd, (int)offsets[l], slot, b.kc(l), b.vc(l),
                             m.opts.kv == PTTS_KV_BF16, b.cap, io);
        PTTS_HIP(hipStreamSynchronize(io));
    }
    b.kv_len_host[slot] = (int32_t)offsets[0];
    b.kv_bound = std::max(b.kv_bound, (int)offsets[0]);
    b.pre_len_host[slot] = 0;
    if (b.slot_local) return;
    h2d(b.st.kv_len + slot, &b.kv_len_host[slot], sizeof(int32_t), io);
    h2d(b.pre_len.as<int32_t>() + slot, &b.pre_len_host[slot], sizeof(int32_t), io);
}

// FlowLM.PromptText -> flowTransformer.prefill (flow_lm.go:155-187, flow_transformer.go:749-771), all slots at once,
// ragged prompts packed as rows.  The hidden output is discarded by the reference, so the last layer stops after
// its keys/values are in the cache.
static int pick_split(int M, int N, int K, bool w_bf16 = false);

void batch_prompt(Batch& b, const float* rows_dev, const int64_t* row_offsets) {
    Model& m = *b.m;
    const Desc& d = m.d;
    hipStream_t s = b.io_stream ? b.io_stream : m.stream;
    const int B = b.B, D = d.d_model;
    const int64_t R64 = row_offsets[B];
    if (R64 == 0) return;
    if (R64 < 0 || R64 > (1 << 24)) throw Error(PTTS_EINVAL, "ptts-hip: prompt row count out of range");
    const int R = (int)R64;
    std::vector<int32_t> row_slot((size_t)R), row_pos((size_t)R);
    int max_pos = 0;
    for (int sl = 0; sl < B; sl++) {
        int64_t t = row_offsets[sl + 1] - row_offsets[sl];
        if (t < 0) throw Error(PTTS_EINVAL, "ptts-hip: prompt row offsets must be non-decreasing");
        if (b.kv_len_host[sl] + t > b.cap) throw Error(PTTS_EINVAL, strfmt("ptts-hip: prompt of %lld rows does not fit KV capacity %d (offset %d)", (long long)t, b.cap, b.kv_len_host[sl]));
        for (int64_t i = 0; i < t; i++) {
            int r = (int)(row_offsets[sl] + i);
            row_slot[r] = sl;
            row_pos[r] = b.kv_len_host[sl] + (int)i;
            max_pos = std::max(max_pos, row_pos[r]);
        }
    }
    const size_t f = sizeof(float);
    // one upload: row -> slot, row -> position, and per slot the first row / first position (the ragged-segment view of the same)
    std::vector<int32_t> meta((size_t)2 * R + 2 * B + 1);
    std::copy(row_slot.begin(), row_slot.end(), meta.begin());
    std::copy(row_pos.begin(), row_pos.end(), meta.begin() + R);
    int max_rows = 0;
    for (int sl = 0; sl <= B; sl++) meta[(size_t)2 * R + sl] = (int32_t)row_offsets[sl];
    for (int sl = 0; sl < B; sl++) {
        meta[(size_t)2 * R + B + 1 + sl] = b.kv_len_host[sl];
        max_rows = std::max(max_rows, (int)(row_offsets[sl + 1] - row_offsets[sl]));
    }
    DevBuf& idx = m.work(2, meta.size() * sizeof(int32_t));
    int32_t* d_slot = idx.as<int32_t>();
    int32_t* d_pos = d_slot + R;
    const int32_t* d_off = d_pos + R;
    const int32_t* d_pos0 = d_off + B + 1;
    h2d(d_slot, meta.data(), meta.size() * sizeof(int32_t), s);
    DevBuf& wsb = m.work(3, ((size_t)R * (size_t)(D + D + 3 * D + D + d.ffn)) * f);
    float* px = wsb.as<float>();
    float* pxn = px + (size_t)R * D;
    float* pqkv = pxn + (size_t)R * D;
    float* pattn = pqkv + (size_t)R * 3 * D;
    float* pff = pattn + (size_t)R * D;
    PTTS_HIP(hipMemcpyAsync(px, rows_dev, (size_t)R * D * f, hipMemcpyDeviceToDevice, s));
    const bool kvb = m.opts.kv == PTTS_KV_BF16;
    // a short prompt at a small batch (R <= 64 rows): linear2 (K = 4 x d_model) runs as the step's split-K kernel and its
    // partial sums are added by the next layer's LayerNorm launch, like in the AR step; otherwise it is one tile GEMM
    const float* pend_partial = nullptr;
    const float* pend_bias = nullptr;
    int pend_split = 0;
    for (int l = 0; l < d.n_layers; l++) {
        const auto& L = d.layers[l];
        {
            LnArgs ln = mkln(m, px, flat(D), L.n1, pxn, D, R);
            ln.partial = pend_partial; ln.splitk = pend_split; ln.pstride = (int64_t)R * D; ln.pbias = pend_bias;
            launch_layernorm(ln, s);
            pend_partial = nullptr; pend_split = 0;
        }
        {
            GemmArgs gq = mk(m, pxn, flat(D), L.in_proj, pqkv, flat(3 * D), R);
            gq.rope_cos = m.at<float>(d.rope_cos); gq.rope_sin = m.at<float>(d.rope_sin);   // q and k are rotated in the epilogue
            gq.rope_cols = 2 * D; gq.rope_hd = d.hd; gq.rope_row_pos = d_pos;
            if (!launch_gemm_rope(gq, s)) {
                gq.rope_cos = gq.rope_sin = nullptr; gq.rope_row_pos = nullptr;
                launch_gemm(gq, s);
                launch_rope_rows(pqkv, flat(3 * D), 0, d.heads, d.hd, d_pos, 0, 0, R, m.at<float>(d.rope_cos), m.at<float>(d.rope_sin), s);
                launch_rope_rows(pqkv, flat(3 * D), D, d.heads, d.hd, d_pos, 0, 0, R, m.at<float>(d.rope_cos), m.at<float>(d.rope_sin), s);
            }
        }
        launch_kv_append(pqkv, 3 * D, D, d.heads, d.hd, d_slot, d_pos, R, b.kc(l), b.vc(l), kvb, b.cap, s);
        if (l == d.n_layers - 1) break;
        AttnArgs a;
        a.q = pqkv; a.q_ld = 3 * D; a.q_col0 = 0;
        a.k = b.kc(l); a.v = b.vc(l); a.kv_bf16 = kvb;
        a.k_seg_stride = (int64_t)d.heads * b.cap * d.hd; a.k_head_stride = (int64_t)b.cap * d.hd; a.k_row_stride = d.hd;
        a.row_seg = d_slot; a.row_pos = d_pos;
        a.rag_off = d_off; a.rag_pos0 = d_pos0; a.rag_segs = B; a.rows_per_seg = max_rows;
        a.context = -1;
        a.out = pattn; a.out_ld = D;
        a.rows = R; a.heads = d.heads; a.max_keys = max_pos + 1;
        launch_attention(a, s);
        GemmArgs go = mk(m, pattn, flat(D), L.out_proj, px, flat(D), R);
        go.R = px; go.epi = EPI_RESADD;
        launch_gemm(go, s);
        launch_layernorm(mkln(m, px, flat(D), L.n2, pxn, D, R), s);
        GemmArgs g1 = mk(m, pxn, flat(D), L.l1, pff, flat(d.ffn), R);
        g1.epi = EPI_GELU;
        launch_gemm(g1, s);
        GemmArgs g2 = mk(m, pff, flat(d.ffn), L.l2, px, flat(D), R);
        const int S = R <= kSkinnyChunkRows ? pick_split(std::min(R, 64), D, d.ffn) : 1;
        GemmArgs g2c = g2;
        g2c.M = std::min(R, 64);
        if (S > 1 && skinny_supported(g2c, S)) {   // (64-row chunks of the split step kernel; the planes of all chunks form one [S][R][D] operand)
            DevBuf& pb = m.work(9, (size_t)S * R * D * f);
            for (int r0 = 0; r0 < R; r0 += 64) {
                g2c = g2;
                g2c.M = std::min(64, R - r0);
                g2c.A = g2.A + (int64_t)r0 * d.ffn;
                g2c.zstride = (int64_t)R * D;
                launch_skinny(g2c, SkinnyFuse{}, S, pb.as<float>() + (size_t)r0 * D, s);
            }
            pend_partial = pb.as<float>(); pend_split = S; pend_bias = m.at<float>(L.l2.b);
        } else {
            // many rows, few row panels: a 4 d_model-deep product on ~100-200 blocks leaves one wave per SIMD waiting on its own
            // loads -- cut K in four (four times the blocks) and let the next layer's LayerNorm launch add the planes
            GemmArgs gs = g2;
            gs.bias = nullptr; gs.kslice = 1024; gs.zstride = (int64_t)R * D;
            const int Sg = (d.ffn + gs.kslice - 1) / gs.kslice;
            if (R < 16384 && Sg > 1 && Sg <= 8) {
                DevBuf& pb = m.work(9, (size_t)Sg * R * D * f);
                gs.C = pb.as<float>(); gs.cmap = flat(D);
                if (gemm5_supported(gs) || gemm3_supported(gs)) {
                    if (gemm5_supported(gs)) launch_gemm5(gs, s);   // (bf16 weights from 1024 rows; the same k order within a slice: the same planes)
                    else launch_gemm3(gs, s);
                    pend_partial = pb.as<float>(); pend_split = Sg; pend_bias = m.at<float>(L.l2.b);
                    continue;
                }
            }
            g2.R = px; g2.epi = EPI_RESADD;
            launch_gemm(g2, s);
        }
    }
    for (int sl = 0; sl < B; sl++) {
        b.kv_len_host[sl] += (int32_t)(row_offsets[sl + 1] - row_offsets[sl]);
        b.kv_bound = std::max(b.kv_bound, (int)b.kv_len_host[sl]);
    }
    if (!b.slot_local) h2d(b.st.kv_len, b.kv_len_host.data(), (size_t)B * sizeof(int32_t), s);
}

// every linear of the AR step goes through here so that bench.py can time the dominant kernel with HIP events
static void step_gemm(Model& m, const GemmArgs& g, const SkinnyFuse& fu = SkinnyFuse{}, int splitk = 1, float* partial = nullptr, hipStream_t st = nullptr) {
    if (!st) st = m.stream;
    const bool fused = fu.partial || fu.ln;
    const bool sk = fused ? skinny_fuse_supported(g, fu) : skinny_supported(g, splitk);
    if (!sk && (splitk > 1 || fused)) throw Error(PTTS_EINVAL, "ptts-hip: internal: shape not supported by the step kernel");
    Prof& p = m.prof;
    if (p.on) {
        while (p.ev.size() < p.used + 2) { hipEvent_t e; PTTS_HIP(hipEventCreate(&e)); p.ev.push_back(e); }
        if (sk) { g_skinny_ev[0] = p.ev[p.used]; g_skinny_ev[1] = p.ev[p.used + 1]; }   // the dispatch stamps itself
        else PTTS_HIP(hipEventRecord(p.ev[p.used], st));
    }
    if (sk) launch_skinny(g, fu, splitk, partial, st);
    else launch_gemm(g, st);
    g_skinny_ev[0] = g_skinny_ev[1] = nullptr;
    if (p.on) {
        if (!sk) PTTS_HIP(hipEventRecord(p.ev[p.used + 1], st));
        p.used += 2;
        p.launches++;
        const int wbytes = (sk && g.wt_i8) ? 1 : (g.w_bf16 ? 2 : 4);
        p.bytes += (double)g.N * g.K * wbytes + (double)g.M * g.K * 4 * (1 + fu.psplit) + (double)g.M * g.N * 4 * (splitk > 1 ? splitk : 1);
        p.wbytes += (double)g.N * g.K * wbytes;
    }
}

// deferred "x += gate * (sum of split-K partials + bias)": executed by the prologue of the next consumer of x
struct Pending {
    const float* partial = nullptr;
    int splitk = 0;
    int64_t pstride = 0;
    const float* bias = nullptr;
};

static int pick_split(int M, int N, int K, bool w_bf16) {
    if (K <= 1024) return 1;   // a whole-K block is one memory burst; splitting pays only when K forces several bursts
    // a full batch on bf16 weights: 2048-deep slices (k_skinny NJ = 8, 32-column blocks): the producer's blocks fetch a third more,
    // every consumer block of the next launch re-reads half as many planes
    if (w_bf16 && M > 32 && K >= 4096 && K % 2048 == 0 && ((N + 31) / 32) * ((M + 15) / 16) * (K / 2048) >= 200) return K / 2048;
    int need = (K + 1023) / 1024;
    int blocks = ((N + 63) / 64) * ((M + 15) / 16);
    int want = (256 + blocks - 1) / blocks;
    return std::max(need, std::min(want, K / 512));
}

// one transformer-style block input: rows of x (with a pending residual update) -> LayerNorm -> linear
struct FusedIn {
    Pending pend;
    float* x_out = nullptr;   // where the updated residual goes when pend is set
    const Norm* norm = nullptr; bool affine = true; float eps = 1e-5f;
    const float* shift = nullptr; const float* scale = nullptr; int64_t ldmod = 0;
    float* y_out = nullptr;
    const StepFinish* finish = nullptr;   // the step's last launch: bookkeeping in the epilogue (SkinnyFuse::fin)
    bool chain = false; const float* chain_noise = nullptr; int64_t chain_noise_stride = 0;   // ... and the next step's opening (SkinnyFuse::chain)
};

// returns whether in.finish was honoured (false: the caller launches k_step_finish itself)
static bool step_fused_linear(Batch& b, const float* x, const FusedIn& in, const Lin& l, float* C, int64_t ldc, int M, int epi,
                              const float* addvec, const float* R, float alpha, hipStream_t st = nullptr, bool* chained = nullptr) {
    Model& m = *b.m;
    GemmArgs g = mk(m, x, flat(l.in), l, C, flat(ldc), M);
    g.epi = epi; g.addvec = addvec; g.R = R; g.alpha = alpha;
    SkinnyFuse fu;
    fu.partial = in.pend.partial; fu.psplit = in.pend.splitk; fu.pstride = in.pend.pstride; fu.pbias = in.pend.bias;
    fu.x_out = in.x_out;
    fu.ln = 1; fu.eps = in.eps;
    if (in.norm && in.affine) { fu.ln_w = m.at<float>(in.norm->w); fu.ln_b = m.at<float>(in.norm->b); }
    fu.shift = in.shift; fu.scale = in.scale; fu.ldmod = in.ldmod;
    fu.y_out = in.y_out;
    fu.fin = in.finish;
    if (in.finish && in.chain) {
        fu.chain = 1; fu.chain_noise = in.chain_noise; fu.chain_noise_stride = in.chain_noise_stride;
        if (skinny_fuse_supported(g, fu)) { step_gemm(m, g, fu, 1, nullptr, st); if (chained) *chained = true; return true; }
        fu.chain = 0; fu.chain_noise = nullptr;
    }
    if (skinny_fuse_supported(g, fu)) { step_gemm(m, g, fu, 1, nullptr, st); return fu.fin != nullptr; }
    fu.fin = nullptr;
    if (skinny_fuse_supported(g, fu)) { step_gemm(m, g, fu, 1, nullptr, st); return false; }
    if (st && st != m.stream) throw Error(PTTS_EINVAL, "ptts-hip: internal: unfused step linear on a side stream");
    // shapes outside the fused kernel (never the reference checkpoint): separate LayerNorm launch, then the linear
    LnArgs ln;
    ln.x = x; ln.xmap = flat(l.in); ln.eps = in.eps; ln.rows = M; ln.d = l.in;
    if (in.norm && in.affine) { ln.w = m.at<float>(in.norm->w); ln.b = m.at<float>(in.norm->b); }
    ln.shift = in.shift; ln.scale = in.scale; ln.ldmod = in.ldmod;
    ln.partial = in.pend.partial; ln.splitk = in.pend.splitk; ln.pstride = in.pend.pstride; ln.pbias = in.pend.bias;
    float* y = in.y_out ? in.y_out : b.xn.as<float>();
    ln.y = y; ln.ldy = l.in;
    if (in.pend.partial) {   // the unfused reducer updates x in place; mirror it into x_out for the caller's ping-pong
        launch_layernorm(ln, m.stream);
        if (in.x_out && in.x_out != x) PTTS_HIP(hipMemcpyAsync(in.x_out, x, (size_t)M * l.in * sizeof(float), hipMemcpyDeviceToDevice, m.stream));
    } else launch_layernorm(ln, m.stream);
    GemmArgs g2 = mk(m, y, flat(l.in), l, C, flat(ldc), M);
    g2.epi = epi; g2.addvec = addvec; g2.R = R; g2.alpha = alpha;
    step_gemm(m, g2);
    return false;
}

// FlowLM.SampleNextLatentStateful minus the host glue (flow_lm.go:252-288): input_linear, 6 x forwardWithState(Tq=1),
// out_norm, out_eos, LSD decode through the flow net.  Everything reads device state, so the sequence is graph-capturable.
// Launch plan per transformer layer: [LN1 + in_proj] -> [RoPE + KV append + attention] -> [out_proj + residual]
// -> [LN2 + linear1 + GELU] -> [linear2, split over K]; the split-K sum and its residual add ride in the next
// layer's first launch.  The residual stream ping-pongs between two buffers so that no launch reads rows another
// block of the same launch rewrites.
// the two ldim-wide linears at the head of a step, as the opening kernels take them; false: shapes they do not take (step_core then runs them as step linears)
static bool open_linears(Batch& b, StepOpenLinears& lin) {
    Model& m = *b.m;
    const Desc& d = m.d;
    const int ld = d.ldim;
    if (!(ld <= 64 && ld % 4 == 0 && d.input_linear.in == ld && d.input_proj.in == ld && d.input_linear.bf16 == d.input_proj.bf16)) return false;
    lin.w_in = m.arena + d.input_linear.w; lin.b_in = m.at<float>(d.input_linear.b); lin.d_in = d.input_linear.out; lin.x = b.x.as<float>();
    lin.w_pj = m.arena + d.input_proj.w; lin.b_pj = m.at<float>(d.input_proj.b); lin.d_pj = d.input_proj.out; lin.fx = b.fx.as<float>();
    lin.w_bf16 = d.input_linear.bf16;
    return true;
}

// a hand-off inside k_flow_cluster gave up (flow_cluster.hip: bounded sweeps -- a tile's eight workgroups were not all running: another tenant on the CUs, a
// masked queue): the frames computed since are not to be trusted.  The exchange state is cleared, the batch -- and whatever batch this engine builds
// later -- goes back to the 2 x depth launches, which compute the same bits, and the event is counted (ptts_dispatch_stats.flow_cluster_fallbacks).
void flow_cluster_recover(Batch& b) {
    hipStream_t s = b.m->stream;
    (void)hipStreamSynchronize(s);
    (void)hipMemsetAsync(b.fc_xbuf.p, 0, flow_cluster_xbuf_bytes(b.B), s);
    (void)hipMemsetAsync(b.fc_sync.p, 0, kFlowClusterSyncBytes, s);
    (void)hipStreamSynchronize(s);
    b.fc_ok = false;
    // (the captured step graphs hold the cluster launch: drop them, the next replay captures the launches)
    for (auto& set : b.graphs) for (auto& row : set) for (hipGraphExec_t& g : row) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    for (auto& gs : b.graph_steps) gs[0] = gs[1] = 0;
    b.opened = false;
    b.m->fc_disabled.store(true);
    b.m->fc_fallbacks.fetch_add(1);
}
void flow_cluster_fault(Batch& b) {
    flow_cluster_recover(b);
    throw FlowClusterFault(PTTS_ENODEVICE, "ptts-hip: the flow net's in-launch hand-off timed out (k_flow_cluster); the steps concerned are re-issued as launches");
}

void step_open(Batch& b) {
    Model& m = *b.m;
    const Desc& d = m.d;
    const int ld = d.ldim;
    const int64_t ls = (int64_t)b.max_steps * ld;
    StepOpenLinears lin;
    const bool ok = open_linears(b, lin);
    b.opened = ok;
    b.par = 0;   // k_step_begin writes the first pair (fx, cur)
    launch_step_begin(b.st, b.latents.as<float>(), ls, m.at<float>(d.bos), b.has_noise ? b.noise.as<float>() : nullptr, ls, ld, b.B,
                      b.in32.as<float>(), b.cur.as<float>(), ok ? &lin : nullptr, m.stream);
}

bool step_core(Batch& b, int lsd, bool opened, bool fuse_finish, bool chain) {
    bool finished = false;
    b.opened = false;   // x / fx are consumed by this step; a chained last launch (below) sets it again for the next one
    if (!opened) b.par = 0;   // (the staged entry points and shapes k_step_begin does not take: the caller filled in32 / cur)
    Model& m = *b.m;
    const Desc& d = m.d;
    hipStream_t s = m.stream;
    const int B = b.B, D = d.d_model, C = d.flow_dim;
    const bool kvb = m.opts.kv == PTTS_KV_BF16;
    float* x = b.x.as<float>();
    float* qkv = b.qkv.as<float>();
    float* attn = b.attn.as<float>();
    float* ff = b.ff.as<float>();
    if (!opened) step_gemm(m, mk(m, b.in32.as<float>(), flat(d.ldim), d.input_linear, x, flat(D), B));
    // A split linear2 leaves [x + (sums_0 + bias)] in plane 0 and the raw sums of the other K slices in planes 1..S-1 (k_skinny, slice 0
    // with GemmArgs::R set): the next consumer of the residual stream reads plane 0 as its rows and adds the others in order --
    // with S = 2 (a full batch on bf16 weights) two row images per block instead of three (residual, two planes) plus the bias.
    Pending pend;
    const float* xin = x;   // where the current residual rows are read from: x, or plane 0 while a split sum is pending
    // 128 rows and more (tall.hip): the prologue once per row (k_rowprep -> bf16 row planes), in_proj / linear1 / linear2 as 64 x 64-tile products (k_tall)
    const bool tall = b.tall_ok && B >= kTallMinRows;
    uint16_t* pa_h = tall ? b.tp_a.as<uint16_t>() : nullptr; uint16_t* pa_l = tall ? pa_h + (size_t)B * D : nullptr;
    uint16_t* pf_h = tall ? b.tp_f.as<uint16_t>() : nullptr; uint16_t* pf_l = tall ? pf_h + (size_t)B * d.ffn : nullptr;
    auto prep = [&](const float* rows, const Pending& pd, const Norm& nm) {
        PrepArgs pa;
        pa.x = rows; pa.ldx = D; pa.partial = pd.partial; pa.psplit = pd.splitk; pa.pstride = pd.pstride; pa.pbias = pd.bias;
        pa.x_out = pd.partial ? x : nullptr;
        pa.ln_w = m.at<float>(nm.w); pa.ln_b = m.at<float>(nm.b); pa.eps = nm.eps;
        pa.yh = pa_h; pa.yl = pa_l; pa.ldy = D; pa.M = B; pa.D = D;
        if (!rowprep_supported(pa)) throw Error(PTTS_EINVAL, "ptts-hip: internal: shape not supported by the row preparation kernel");
        launch_rowprep(pa, s);
    };
    auto tall_lin = [&](const uint16_t* ah, const uint16_t* al, const Lin& li, TallArgs t) {
        t.ah = ah; t.al = al; t.lda = li.in; t.Wt = m.arena + li.wt; t.bias = m.at<float>(li.b); t.M = B; t.N = li.out; t.K = li.in;
        if (!tall_supported(t)) throw Error(PTTS_EINVAL, "ptts-hip: internal: shape not supported by the 64-row step linear");
        launch_tall(t, s);
    };
    for (int l = 0; l < d.n_layers; l++) {
        const auto& L = d.layers[l];
        if (tall) {
            prep(xin, pend, L.n1);
            pend = Pending{}; xin = x;
            TallArgs t; t.C = qkv; t.ldc = 3 * D;
            tall_lin(pa_h, pa_l, L.in_proj, t);
        } else {
            FusedIn in;
            in.pend = pend; in.x_out = pend.partial ? x : nullptr; in.norm = &L.n1; in.eps = L.n1.eps;
            step_fused_linear(b, xin, in, L.in_proj, qkv, 3 * D, B, EPI_NONE, nullptr, nullptr, 1.0f);
            pend = Pending{}; xin = x;
        }
        AttnArgs a;
        a.k = b.kc(l); a.v = b.vc(l); a.kv_bf16 = kvb;
        a.k_seg_stride = (int64_t)d.heads * b.cap * d.hd; a.k_head_stride = (int64_t)b.cap * d.hd; a.k_row_stride = d.hd;
        a.seg_len = b.st.kv_len; a.active = b.st.active;
        a.context = -1;
        a.out = attn; a.out_ld = D;
        a.rows = B; a.heads = d.heads; a.max_keys = b.cap;
        a.keys_now = b.capturing ? b.capture_keys : b.kv_bound + 1;
        a.fused_step = 1; a.qkv = qkv; a.qkv_ld = 3 * D; a.d_model = D;
        a.cos_t = m.at<float>(d.rope_cos); a.sin_t = m.at<float>(d.rope_sin); a.cap = b.cap;
        a.pre_k = b.pre_k.as<const void*>(); a.pre_v = b.pre_v.as<const void*>(); a.pre_len = b.pre_len.as<int32_t>(); a.layer = l;
        launch_attention(a, s);
        {
            GemmArgs go = mk(m, attn, flat(D), L.out_proj, x, flat(D), B);
            go.R = x; go.epi = EPI_RESADD;
            step_gemm(m, go);
        }
        if (tall) {
            prep(x, Pending{}, L.n2);
            TallArgs t1; t1.ch = pf_h; t1.cl = pf_l; t1.ldp = d.ffn; t1.epi = EPI_GELU;   // gelu(linear1) leaves as the planes linear2 reads
            tall_lin(pa_h, pa_l, L.l1, t1);
            const int S = std::max(1, std::min(4, d.ffn / 1024));   // 1024-deep K slices: 16 x rows / 64 x S blocks of 8 chunks
            float* planes = b.partial.as<float>();
            TallArgs t2; t2.R = x; t2.ldr = D;
            if (S > 1) { t2.splitk = S; t2.partial = planes; t2.zstride = (int64_t)B * D; }
            else { t2.epi = EPI_RESADD; t2.C = x; t2.ldc = D; }
            tall_lin(pf_h, pf_l, L.l2, t2);
            if (S > 1) { xin = planes; pend.partial = planes + (size_t)B * D; pend.splitk = S - 1; pend.pstride = (int64_t)B * D; pend.bias = nullptr; }
            continue;
        }
        {
            FusedIn in;
            in.norm = &L.n2; in.eps = L.n2.eps;
            step_fused_linear(b, x, in, L.l1, ff, d.ffn, B, EPI_GELU, nullptr, nullptr, 1.0f);
        }
        {
            GemmArgs g2 = mk(m, ff, flat(d.ffn), L.l2, x, flat(D), B);
            const int S = pick_split(B, D, d.ffn, g2.w_bf16 != 0 || g2.wt_i8 != 0);
            if (S > 1 && skinny_supported(g2, S)) {
                float* planes = b.partial.as<float>();
                g2.R = x;   // slice 0 stores x + (sums_0 + bias)
                step_gemm(m, g2, SkinnyFuse{}, S, planes);
                xin = planes;
                pend.partial = planes + (size_t)B * D; pend.splitk = S - 1; pend.pstride = (int64_t)B * D; pend.bias = nullptr;
            } else {
                g2.R = x; g2.epi = EPI_RESADD;
                step_gemm(m, g2);
            }
        }
    }
    float* last = b.last.as<float>();
    float* sy = b.sy.as<float>();
    const float* tc = m.tcomb.at(lsd)->as<float>();
    // out_norm (flow_lm.go:262) feeds out_eos and the flow net's cond_embed.  Both linears carry the norm (and the pending
    // split-K sum of the last linear2) in their prologue -- one launch fewer than a stand-alone LayerNorm (parallel graph
    // branches were tried for these two and for input_proj: the multi-branch graph ran the step 3x slower on ROCm 7.2);
    // cond_embed also writes the normalised rows
    // (`last`: later Euler steps and the staged API read them).  Shapes the fused kernel does not take keep the
    // stand-alone LayerNorm launch.
    bool tail_fused = false;
    {
        FusedIn in;
        in.pend = pend; in.norm = &d.out_norm; in.eps = d.out_norm.eps;
        GemmArgs g1 = mk(m, xin, flat(D), d.cond_embed, sy, flat(C), B), g2 = mk(m, xin, flat(D), d.out_eos, b.eos.as<float>(), flat(1), B);
        SkinnyFuse fu;
        fu.partial = pend.partial; fu.psplit = pend.splitk; fu.pstride = pend.pstride; fu.pbias = pend.bias;
        fu.ln = 1; fu.eps = in.eps; fu.ln_w = m.at<float>(d.out_norm.w); fu.ln_b = m.at<float>(d.out_norm.b);
        tail_fused = skinny_fuse_supported(g1, fu) && skinny_fuse_supported(g2, fu);
        if (tall && d.cond_eos.wt != NONE) {
            // 128 rows and more: the pending sum and out_norm once per row (k_rowprep writes `last`); [cond_embed ; out_eos] then reads finished rows -- every
            // block of the fused form would redo the four-plane prologue (16.3 us at 256 rows against 4.7 + 8)
            PrepArgs pa;
            pa.x = xin; pa.ldx = D; pa.partial = pend.partial; pa.psplit = pend.splitk; pa.pstride = pend.pstride; pa.pbias = pend.bias;
            pa.x_out = pend.partial ? x : nullptr;
            pa.ln_w = m.at<float>(d.out_norm.w); pa.ln_b = m.at<float>(d.out_norm.b); pa.eps = d.out_norm.eps;
            pa.yh = pa_h; pa.yl = pa_l; pa.ldy = D; pa.y_out = last; pa.M = B; pa.D = D;
            if (!rowprep_supported(pa)) throw Error(PTTS_EINVAL, "ptts-hip: internal: shape not supported by the row preparation kernel");
            launch_rowprep(pa, s);
            GemmArgs g = mk(m, last, flat(D), d.cond_eos, sy, flat(C), B);
            g.epi = EPI_SILU; g.addvec = tc; g.tail = b.eos.as<float>();
            step_gemm(m, g);
            tail_fused = true;
        } else if (tail_fused) {
            in.y_out = last;
            if (d.cond_eos.wt != NONE) {   // one launch: columns 0..C-1 = cond_embed (SiLU epilogue), column C = out_eos (raw)
                GemmArgs g = mk(m, xin, flat(D), d.cond_eos, sy, flat(C), B);
                g.epi = EPI_SILU; g.addvec = tc; g.tail = b.eos.as<float>();
                SkinnyFuse f2 = fu;
                f2.y_out = last;
                step_gemm(m, g, f2);
            } else {
                in.y_out = nullptr;
                step_fused_linear(b, xin, in, d.out_eos, b.eos.as<float>(), 1, B, EPI_NONE, nullptr, nullptr, 1.0f);
                in.y_out = last;
                step_fused_linear(b, xin, in, d.cond_embed, sy, C, B, EPI_SILU, tc, nullptr, 1.0f);
            }
        } else {
            LnArgs ln = mkln(m, xin, flat(D), d.out_norm, last, D, B);
            ln.partial = pend.partial; ln.splitk = pend.splitk; ln.pstride = pend.pstride; ln.pbias = pend.bias;
            launch_layernorm(ln, s);
            step_gemm(m, mk(m, last, flat(D), d.out_eos, b.eos.as<float>(), flat(1), B));
        }
    }
    b.tail_fused = tail_fused;
    finished = step_flow(b, lsd, opened, fuse_finish, chain);
    if (!b.capturing) b.kv_bound++;   // every live slot has appended one key
    return finished;
}

// LSDDecode (flow_lm.go:311-353) with flowNet.Forward (flow_net.go:314-356) per Euler step: the flow part of a step, from `last` (out_norm's rows), `sy`
// (when the transformer's last launch produced it: Batch::tail_fused) and `cur` (x0 -> the frame)
static bool step_flow(Batch& b, int lsd, bool opened, bool fuse_finish, bool chain) {
    bool finished = false;
    Model& m = *b.m;
    const Desc& d = m.d;
    const int B = b.B, D = d.d_model, C = d.flow_dim, NA = d.ada_all.out;
    hipStream_t s = m.stream;
    const bool tail_fused = b.tail_fused;
    float* last = b.last.as<float>();
    float* sy = b.sy.as<float>();
    const float* tc = m.tcomb.at(lsd)->as<float>();
    float* ada = b.ada.as<float>();
    float* fx = b.fx_now();
    float* fh2 = b.fh2.as<float>();
    float* cur = b.cur_now();
    for (int i = 0; i < lsd; i++) {
        if (i > 0 || !tail_fused) {
            GemmArgs gc = mk(m, last, flat(D), d.cond_embed, sy, flat(C), B);  // sy = silu(0.5*(e_s+e_t) + cond_embed(c))
            gc.epi = EPI_SILU; gc.addvec = tc + (size_t)i * C;
            step_gemm(m, gc);
        }
        step_gemm(m, mk(m, sy, flat(C), d.ada_all, ada, flat(NA), B));
        if (i > 0 || !opened) step_gemm(m, mk(m, cur, flat(d.ldim), d.input_proj, fx, flat(C), B));
        bool clustered = false;
        if (b.fc_ok) {   // all residual blocks in one launch (flow_cluster.hip)
            FlowClusterArgs fa;
            fa.fx_in = fx; fa.fx_out = fx; fa.ada = ada; fa.ldmod = NA; fa.rows = B; fa.depth = d.flow_depth;
            for (int r = 0; r < d.flow_depth; r++) {
                const auto& rb = d.rb[r];
                fa.eps[r] = rb.ln.eps; fa.ln_w[r] = m.at<float>(rb.ln.w); fa.ln_b[r] = m.at<float>(rb.ln.b);
                fa.w0[r] = m.arena + rb.mlp0.wt; fa.b0[r] = m.at<float>(rb.mlp0.b);
                fa.w2[r] = m.arena + rb.mlp2.wt; fa.b2[r] = m.at<float>(rb.mlp2.b);
            }
            fa.xbuf = b.fc_xbuf.as<unsigned long long>(); fa.sync = b.fc_sync.as<unsigned>();
            if (b.fc_stamps.p) fa.stamps = b.fc_stamps.as<unsigned long long>();
            if (m.fc_inject && !b.capturing) { fa.inject = m.fc_inject; m.fc_inject = 0; }
            if (flow_cluster_supported(fa, C)) {
                // (not a launch of the step linear: the bench's per-launch events and byte counts -- Prof -- leave it out; bench.py reports it from the kernel trace)
                launch_flow_cluster(fa, s);
                clustered = true;
            }
        }
        for (int r = 0; !clustered && r < d.flow_depth; r++) {  // flowResBlock.Forward flow_net.go:116-172 (chunks: shift, scale, gate)
            const auto& rb = d.rb[r];
            FusedIn in;
            in.norm = &rb.ln; in.eps = rb.ln.eps;
            in.shift = ada + (size_t)r * 3 * C; in.scale = in.shift + C; in.ldmod = NA;
            step_fused_linear(b, fx, in, rb.mlp0, fh2, C, B, EPI_SILU, nullptr, nullptr, 1.0f);
            GemmArgs g2 = mk(m, fh2, flat(C), rb.mlp2, fx, flat(C), B);
            g2.R = fx; g2.epi = EPI_GATE_RESADD; g2.gate = ada + (size_t)r * 3 * C + 2 * C; g2.ldg = NA;
            step_gemm(m, g2);
        }
        FusedIn fin;  // flowFinalLayer.Forward flow_net.go:205-239: LayerNorm without affine, eps 1e-6, chunks: shift, scale
        fin.affine = false; fin.eps = 1e-6f;
        fin.shift = ada + (size_t)d.flow_depth * 3 * C; fin.scale = fin.shift + C; fin.ldmod = NA;
        if (fuse_finish && i == lsd - 1) {
            fin.finish = b.fin_dev.as<StepFinish>() + b.par;
            if (chain && b.chain_ok) {   // ... and the NEXT step's opening (k_step_begin's work without its launch)
                fin.chain = true;
                fin.chain_noise = b.has_noise ? b.noise.as<float>() : nullptr;
                fin.chain_noise_stride = (int64_t)b.max_steps * d.ldim;
            }
        }
        bool chained = false;
        finished = step_fused_linear(b, fx, fin, d.final_linear, cur, d.ldim, B, EPI_AXPY, nullptr, cur, 1.0f / (float)lsd, nullptr, &chained);  // current += flow / steps
        if (chained) { b.opened = true; b.par ^= 1; }
    }
    return finished;
}

void step_flow_again(Batch& b, int lsd) {
    b.opened = false; b.par = 0;
    (void)step_flow(b, lsd, false, false, false);
}

// ------------------------------------------------------------------------------------------------
// Mimi: Model.LatentToMimi + MimiModel.DecodeFromLatent (model.go:141-319, mimi.go:719-789)
//
// Every op of the decoder is causal (mimi.go:69-76,116-125,418), and every intermediate lives in a per-utterance
// channels-last buffer that spans the whole utterance.  Decoding frames [f0, f1) is therefore just the same launches on
// a row sub-range of each buffer: history rows (conv taps, the transposed convs' x[t-1], the 250-step attention window)
// are read from what earlier ranges left behind.  generate() uses this to decode finished frames on a second stream
// while the (latency-bound) AR loop is still producing later ones.
// ------------------------------------------------------------------------------------------------
void mimi_setup(Model& m, MimiWs& w, int B, int T) {
    const Desc& d = m.d;
    w.B = B; w.T = T;
    const int C = d.mimi_dim, S = d.up_stride, T1 = T * S, F = d.mimi_ffn;
    w.P0 = d.init_k - 1;
    w.Ls[0] = T1; w.Ps[0] = 1;
    for (int j = 0; j < 3; j++) {
        w.Ls[j + 1] = w.Ls[j] * d.strides[j];
        int need_next = j < 2 ? 1 : d.final_k - 1;
        w.Ps[j + 1] = std::max(d.rb_k1[j] - 1, need_next);
    }
    size_t n_xp = (size_t)B * (1 + T) * C, n_up = (size_t)B * (w.P0 + T1) * C;
    size_t n_n1 = (size_t)B * T1 * C, n_qkv = (size_t)B * T1 * 3 * C, n_ff = (size_t)B * T1 * F;
    size_t n_c[4], n_h[3];
    for (int j = 0; j < 4; j++) n_c[j] = (size_t)B * (w.Ps[j] + w.Ls[j]) * d.sea_ch[j];
    for (int j = 0; j < 3; j++) n_h[j] = (size_t)B * ((d.rb_k2[j] - 1) + w.Ls[j + 1]) * d.sea_hidden[j];
    size_t total = n_xp + n_up + n_n1 + n_qkv * d.mimi_layers + n_n1 + n_ff + n_c[0] + 2 * (n_c[1] + n_c[2] + n_c[3]) + n_h[0] + n_h[1] + n_h[2];
    DevBuf& wsb = m.work(4, total * sizeof(float));
    float* p = wsb.as<float>();
    w.xp = p; p += n_xp;
    w.up = p; p += n_up;
    w.n1 = p; p += n_n1;
    for (int l = 0; l < d.mimi_layers; l++) { w.qkv[l] = p; p += n_qkv; }
    w.attn = p; p += n_n1;
    w.ff = p; p += n_ff;
    w.c0 = p; p += n_c[0];
    for (int j = 0; j < 3; j++) { w.u[j] = p; p += n_c[j + 1]; w.uo[j] = p; p += n_c[j + 1]; w.h[j] = p; p += n_h[j]; }
    w.zeroed = false;
}

void mimi_zero_history(Model& m, MimiWs& w, hipStream_t s) {
    const Desc& d = m.d;
    const int C = d.mimi_dim, T1 = w.Ls[0];
    launch_zero_rows(w.xp, (int64_t)(1 + w.T) * C, w.B, C, s);
    launch_zero_rows(w.up, (int64_t)(w.P0 + T1) * C, w.B, (int64_t)w.P0 * C, s);
    launch_zero_rows(w.c0, (int64_t)(w.Ps[0] + w.Ls[0]) * d.sea_ch[0], w.B, (int64_t)w.Ps[0] * d.sea_ch[0], s);
    for (int j = 0; j < 3; j++) {
        const int cout = d.sea_ch[j + 1], hid = d.sea_hidden[j], Ph = d.rb_k2[j] - 1;
        launch_zero_rows(w.u[j], (int64_t)(w.Ps[j + 1] + w.Ls[j + 1]) * cout, w.B, (int64_t)w.Ps[j + 1] * cout, s);
        launch_zero_rows(w.uo[j], (int64_t)(w.Ps[j + 1] + w.Ls[j + 1]) * cout, w.B, (int64_t)w.Ps[j + 1] * cout, s);
        if (Ph > 0) launch_zero_rows(w.h[j], (int64_t)(Ph + w.Ls[j + 1]) * hid, w.B, (int64_t)Ph * hid, s);
    }
    w.zeroed = true;
}

// Two pieces of a decoder-transformer layer (mimi.go:245-441) as the decoder launches them; also reachable on their own through
// ptts_mimi_layer_piece (staged parity checks).
// norm1 -> in_proj (no bias) -> RoPE of q and k at positions pos0 + (row % rows_per_seg): x rows [R][C] -> qkv rows [R][3C].  n1: [R][C] scratch.
void mimi_layer_qkv(Model& m, int l, const float* x, RowMap xmap, int R, float* qkv, RowMap qmap, int pos0, int rows_per_seg, float* n1, hipStream_t s) {
    const Desc& d = m.d;
    const auto& L = d.ml[l];
    const int C = d.mimi_dim;
    if (L.qkv_img != NONE && d.mimi_hd == 64) {   // one kernel (ffn_fused.hip k_mimi_rowlin)
        RowLinArgs ra;
        ra.x = x; ra.xmap = xmap;
        ra.ln_w = m.at<float>(L.n1.w); ra.ln_b = m.at<float>(L.n1.b); ra.eps = L.n1.eps;
        ra.img = m.at<uint8_t>(L.qkv_img);
        ra.y = qkv; ra.ymap = qmap;
        ra.rope_cos = m.at<float>(d.rope_cos); ra.rope_sin = m.at<float>(d.rope_sin); ra.rope_cols = 2 * C; ra.rope_pos0 = pos0; ra.rope_rows_per_seg = rows_per_seg;
        ra.M = R; ra.N = 3 * C; ra.K = C;
        if (mimi_rowlin_supported(ra)) { launch_mimi_rowlin(ra, s); return; }
    }
    launch_layernorm(mkln(m, x, xmap, L.n1, n1, C, R), s);
    GemmArgs gq = mk(m, n1, flat(C), L.in_proj, qkv, qmap, R);
    gq.rope_cos = m.at<float>(d.rope_cos); gq.rope_sin = m.at<float>(d.rope_sin);   // q and k are rotated in the epilogue
    gq.rope_cols = 2 * C; gq.rope_hd = d.mimi_hd; gq.rope_pos0 = pos0; gq.rope_rows_per_seg = rows_per_seg;
    if (!launch_gemm_rope(gq, s)) {
        gq.rope_cos = gq.rope_sin = nullptr;
        launch_gemm(gq, s);
        launch_rope_rows(qkv, qmap, 0, d.mimi_heads, d.mimi_hd, nullptr, pos0, rows_per_seg, R, m.at<float>(d.rope_cos), m.at<float>(d.rope_sin), s);
        launch_rope_rows(qkv, qmap, C, d.mimi_heads, d.mimi_hd, nullptr, pos0, rows_per_seg, R, m.at<float>(d.rope_cos), m.at<float>(d.rope_sin), s);
    }
}

// x += layer_scale_2 * linear2(gelu(linear1(norm2(x)))) on rows [R][C], in place.  n1: [R][C], ffb: [R][F] scratch (unused by the fused kernel).
void mimi_layer_ffn(Model& m, int l, float* x, RowMap xmap, int R, float* n1, float* ffb, hipStream_t s) {
    const Desc& d = m.d;
    const auto& L = d.ml[l];
    const int C = d.mimi_dim, F = d.mimi_ffn;
    if (L.ffn_img != NONE) {   // one kernel (ffn_fused.hip k_mimi_ffn)
        FfnArgs fa;
        fa.x = x; fa.xmap = xmap;
        fa.ln_w = m.at<float>(L.n2.w); fa.ln_b = m.at<float>(L.n2.b); fa.eps = L.n2.eps;
        fa.img = m.at<uint8_t>(L.ffn_img);
        fa.ls = m.at<float>(L.ls2);
        fa.M = R; fa.D = C; fa.F = F;
        if (mimi_ffn_supported(fa)) { launch_mimi_ffn(fa, s); return; }
    }
    launch_layernorm(mkln(m, x, xmap, L.n2, n1, C, R), s);
    GemmArgs g1 = mk(m, n1, flat(C), L.l1, ffb, flat(F), R);
    g1.epi = EPI_GELU;
    launch_gemm(g1, s);
    GemmArgs g2 = mk(m, ffb, flat(F), L.l2, x, xmap, R);
    g2.R = x; g2.epi = L.ls2 != NONE ? EPI_SCALE_RESADD : EPI_RESADD; g2.scale = m.at<float>(L.ls2);
    launch_gemm(g2, s);
}

// frames [f0, f1) of every utterance; lat: device [B][*][ldim] with lat_bstride elements between utterances;
// pcm: device [B][T * samples_per_frame]; mimi_latent (optional): [B][C][T] (whole range only)
void mimi_range(Model& m, MimiWs& w, const float* lat, int64_t lat_bstride, int f0, int f1, float* pcm, float* mimi_latent, hipStream_t s,
                const PcmRow* pcm_rows, bool* rows_used, float* xformer_out) {
    if (rows_used) *rows_used = false;
    const Desc& d = m.d;
    const int B = w.B, T = w.T, C = d.mimi_dim, S = d.up_stride, T1 = T * S, F = d.mimi_ffn, P0 = w.P0;
    if (f1 <= f0) return;
    if (!w.zeroed) mimi_zero_history(m, w, s);
    const int nf = f1 - f0;
    // K13: latent -> mimi projection rows 1+f0 .. 1+f1 of xp
    launch_projector(lat, lat_bstride, m.at<float>(d.proj_w), m.at<float>(d.proj_b), B, T, f0, f1, d.ldim, C, w.xp, s);
    if (mimi_latent) launch_btc_to_bct(w.xp, 1, B, C, T, mimi_latent, s);
    // K14: depthwise upsample -> rows [f0*S, f1*S) of up
    launch_upsample_depthwise(w.xp, m.at<float>(d.up_w0), m.at<float>(d.up_w1), nullptr, B, T, f0, f1, C, S, w.up, P0, s);
    // decoder transformer (mimi.go:245-441, 506-525): positions restart at 0 for every utterance, window `context`
    const int t0 = f0 * S, CT = nf * S, R = B * CT;
    (void)F;
    const int64_t up_bs = (int64_t)(P0 + T1) * C;
    float* upx = w.up + (size_t)(P0 + t0) * C;            // chunk rows of the residual stream
    const RowMap upm = seg(C, CT, up_bs);
    float* n1 = w.n1;                                       // [R][C] scratch (chunk-local)
    float* attn = w.attn;
    float* ffb = w.ff;
    for (int l = 0; l < d.mimi_layers; l++) {
        const auto& L = d.ml[l];
        float* qkv = w.qkv[l];                              // [B][T1][3C], persists across ranges (keys/values of earlier frames)
        float* qkvx = qkv + (size_t)t0 * 3 * C;
        const RowMap qm = seg(3 * C, CT, (int64_t)T1 * 3 * C);
        mimi_layer_qkv(m, l, upx, upm, R, qkvx, qm, t0, CT, n1, s);
        AttnArgs a;
        a.q = qkvx; a.q_ld = 3 * C; a.q_col0 = 0; a.q_rows_per_batch = CT; a.q_batch_stride = (int64_t)T1 * 3 * C;
        a.k = qkv + C; a.v = qkv + 2 * C; a.kv_bf16 = 0;
        a.k_seg_stride = (int64_t)T1 * 3 * C; a.k_head_stride = d.mimi_hd; a.k_row_stride = 3 * C;
        a.rows_per_seg = CT; a.pos_base = t0;
        a.context = d.mimi_ctx;
        a.out = attn; a.out_ld = C;
        a.rows = R; a.heads = d.mimi_heads; a.max_keys = std::min(T1, d.mimi_ctx);
        launch_attention(a, s);
        GemmArgs go = mk(m, attn, flat(C), L.out_proj, upx, upm, R);
        go.R = upx; go.epi = L.ls1 != NONE ? EPI_SCALE_RESADD : EPI_RESADD; go.scale = m.at<float>(L.ls1);
        launch_gemm(go, s);
        mimi_layer_ffn(m, l, upx, upm, R, n1, ffb, s);
    }
    if (xformer_out)   // staged parity check: the residual stream after the last layer, [B][T1][C] (rows of this range)
        PTTS_HIP(hipMemcpy2DAsync(xformer_out + (size_t)t0 * C, (size_t)T1 * C * sizeof(float), upx, (size_t)up_bs * sizeof(float), (size_t)CT * C * sizeof(float),
                                  (size_t)B, hipMemcpyDeviceToDevice, s));
    // SEANet decoder (mimi.go:740-788): causal convs as GEMMs over contiguous channels-last windows
    int r0 = t0, rn = CT;   // row range at the current rate
    {
        const int ch = d.sea_ch[0];
        GemmArgs g = mk(m, w.up + (size_t)r0 * C, seg(C, rn, up_bs), d.init_conv, w.c0 + (size_t)(w.Ps[0] + r0) * ch,
                        seg(ch, rn, (int64_t)(w.Ps[0] + w.Ls[0]) * ch), B * rn);
        g.epi = EPI_ELU;  // x = elu(initConv(x))
        if (C % 64 == 0 && d.init_conv.in == d.init_k * C) { g.win_taps = d.init_k; g.win_c = C; }   // (k walked channel-block-major: gemm5.hip)
        launch_gemm(g, s);
    }
    bool final_done = false;
    for (int j = 0; j < 3; j++) {
        const int cin = d.sea_ch[j], cout = d.sea_ch[j + 1], st = d.strides[j], hid = d.sea_hidden[j];
        const int Lin_ = w.Ls[j], Lout = w.Ls[j + 1], Pin = w.Ps[j], Pout = w.Ps[j + 1], Ph = d.rb_k2[j] - 1;
        const float* in = j == 0 ? w.c0 : w.uo[j - 1];
        float* u = w.u[j];
        float* uo = w.uo[j];
        float* hb = w.h[j];
        const int64_t in_bs = (int64_t)(Pin + Lin_) * cin, u_bs = (int64_t)(Pout + Lout) * cout, h_bs = (int64_t)(Ph + Lout) * hid;
        // transposed conv: window [x[t-1], x[t]] starts one row before t
        GemmArgs gu = mk(m, in + (size_t)(Pin - 1 + r0) * cin, seg(cin, rn, in_bs), d.up[j],
                         u + (size_t)(Pout + (int64_t)r0 * st) * cout, seg((int64_t)st * cout, rn, u_bs), B * rn);
        // elu(x) precedes every transposed conv: c0 was activated in the initConv epilogue, uo[j-1] in its residual epilogue
        if (j == 2 && d.rb1[j].wf != NONE && d.rb2[j].wf != NONE && d.rb1[j].bf16 && d.rb2[j].bf16 && d.up[j].wf != NONE) {
            // the last stage as ONE kernel: transposed convolution + residual block + final convolution (resblock_up.hip); u[2] is never written
            ResArgs rr;
            rr.u = u; rr.u_bs = u_bs; rr.pad = Pout; rr.uo = uo;
            rr.w1 = m.at<uint8_t>(d.rb1[j].wf); rr.b1 = m.at<float>(d.rb1[j].b);
            rr.w2 = m.at<uint8_t>(d.rb2[j].wf); rr.b2 = m.at<float>(d.rb2[j].b);
            rr.B = B; rr.L = Lout; rr.t0 = r0 * st; rr.t1 = (r0 + rn) * st;
            rr.C = cout; rr.H = hid; rr.k1 = d.rb_k1[j]; rr.k2 = d.rb_k2[j]; rr.w_bf16 = 1;
            rr.final_conv = 1; rr.kf = d.final_k; rr.wf_hi = m.at<uint8_t>(d.final_wf); rr.wf_lo = m.at<uint8_t>(d.final_wf_lo); rr.bf = m.at<float>(d.final_b);
            rr.pcm = pcm; rr.pcm_bs = w.Ls[3]; rr.pcm_rows = pcm_rows;
            rr.fuse_up = 1; rr.xin = in; rr.x_bs = in_bs; rr.x_pad = Pin; rr.x_L = Lin_; rr.CI = cin; rr.up_stride = st;
            rr.wup = m.at<uint8_t>(d.up[j].wf); rr.bup = m.at<float>(d.up[j].b);
            if (resblock_up_supported(rr)) {
                launch_resblock_up(rr, s);
                if (rows_used && pcm_rows) *rows_used = true;
                final_done = true;
                r0 *= st; rn *= st;
                continue;
            }
        }
        launch_gemm(gu, s);
        r0 *= st; rn *= st;
        // residual block: x + conv_k1(elu(conv_k3(elu(x))))  (mimi.go:146-164); x stays in u, elu(sum) goes to uo -- the
        // only readers of the sum are the next transposed conv and the final conv, both behind an ELU (mimi.go:752-783).
        // The two narrow blocks run as one launch each (resblock.hip); the last one also applies the final conv.
        {
            ResArgs ra;
            ra.u = u; ra.u_bs = u_bs; ra.pad = Pout; ra.uo = uo;
            ra.w1 = m.at<uint8_t>(d.rb1[j].wf); ra.w1_lo = m.at<uint8_t>(d.rb1[j].wf_lo); ra.b1 = m.at<float>(d.rb1[j].b);
            ra.w2 = m.at<uint8_t>(d.rb2[j].wf); ra.w2_lo = m.at<uint8_t>(d.rb2[j].wf_lo); ra.b2 = m.at<float>(d.rb2[j].b);
            ra.B = B; ra.L = Lout; ra.t0 = r0; ra.t1 = r0 + rn;
            ra.C = cout; ra.H = hid; ra.k1 = d.rb_k1[j]; ra.k2 = d.rb_k2[j]; ra.w_bf16 = d.rb1[j].bf16;
            if (j == 2) {
                ra.final_conv = 1; ra.kf = d.final_k; ra.wf_hi = m.at<uint8_t>(d.final_wf); ra.wf_lo = m.at<uint8_t>(d.final_wf_lo); ra.bf = m.at<float>(d.final_b);
                ra.pcm = pcm; ra.pcm_bs = w.Ls[3];
            }
            if (j == 2 && pcm_rows && d.rb1[j].wf != NONE && d.rb2[j].wf != NONE && d.rb1[j].bf16 == d.rb2[j].bf16) {
                ResArgs rr = ra;
                rr.pcm_rows = pcm_rows;
                if (resblock_supported(rr)) {
                    launch_resblock(rr, s);
                    if (rows_used) *rows_used = true;
                    final_done = true;
                    continue;
                }
            }
            if (d.rb1[j].wf != NONE && d.rb2[j].wf != NONE && d.rb1[j].bf16 == d.rb2[j].bf16 && resblock_supported(ra)) {
                launch_resblock(ra, s);
                if (j == 2) final_done = true;
                continue;
            }
        }
        GemmArgs g1 = mk(m, u + (size_t)(Pout - (d.rb_k1[j] - 1) + r0) * cout, seg(cout, rn, u_bs), d.rb1[j],
                         hb + (size_t)(Ph + r0) * hid, seg(hid, rn, h_bs), B * rn);
        g1.aop = AOP_ELU; g1.epi = EPI_ELU;
        if (cout % 64 == 0 && d.rb1[j].in == d.rb_k1[j] * cout) { g1.win_taps = d.rb_k1[j]; g1.win_c = cout; }
        launch_gemm(g1, s);
        GemmArgs g2 = mk(m, hb + (size_t)r0 * hid, seg(hid, rn, h_bs), d.rb2[j], uo + (size_t)(Pout + r0) * cout, seg(cout, rn, u_bs), B * rn);
        g2.R = u + (size_t)(Pout + r0) * cout; g2.epi = EPI_RESADD_ELU;
        launch_gemm(g2, s);
    }
    if (!final_done)
    launch_conv_final(w.uo[2], w.Ps[3], m.at<float>(d.final_w), m.at<float>(d.final_b), B, w.Ls[3], r0, r0 + rn, d.sea_ch[3], d.final_k, 0, pcm, s);
}

void mimi_decode(Model& m, const float* lat, int64_t lat_bstride, int B, int T, float* pcm, float* mimi_latent, float* xformer_out) {
    if (B <= 0 || T <= 0) return;
    if ((int64_t)T * m.d.up_stride > ROPE_SEQ) throw Error(PTTS_EINVAL, strfmt("ops: rope cos/sin sequence length too small for pos=0 seq=%lld", (long long)T * m.d.up_stride));
    // bound the workspace: ~2.7 MB of f32 activations per latent frame at the reference shapes
    const Desc& d = m.d;
    double per_frame = 4.0 * ((double)d.up_stride * (d.mimi_dim * (4.0 + 3.0 * d.mimi_layers) + d.mimi_ffn) +
                              (double)d.samples_per_frame * (d.sea_ch[3] * 2.6 + d.sea_ch[2] * 0.7 + d.sea_ch[1] * 0.25));
    int group = (int)std::max(1.0, std::min((double)B, 40e9 / (per_frame * T)));
    const int64_t spu = (int64_t)T * d.samples_per_frame;
    for (int b0 = 0; b0 < B; b0 += group) {
        int nb = std::min(group, B - b0);
        MimiWs w;
        mimi_setup(m, w, nb, T);
        mimi_range(m, w, lat + (int64_t)b0 * lat_bstride, lat_bstride, 0, T, pcm + (int64_t)b0 * spu,
                   mimi_latent ? mimi_latent + (int64_t)b0 * d.mimi_dim * T : nullptr, m.stream, nullptr, nullptr,
                   xformer_out ? xformer_out + (int64_t)b0 * T * d.up_stride * d.mimi_dim : nullptr);
    }
}


// ------------------------------------------------------------------------------------------------
// pinned result pool
// ------------------------------------------------------------------------------------------------
namespace {
struct PinnedPool {
    std::mutex mu;
    std::map<void*, size_t> owned;                    // every live pinned block -> its (rounded) size
    std::multimap<size_t, void*> free_blocks;         // size -> block
    size_t free_bytes = 0;
    static constexpr size_t kKeep = (size_t)1 << 30;  // keep at most 1 GiB of idle pinned memory
};
PinnedPool& pool() { static PinnedPool* p = new PinnedPool(); return *p; }   // leaked on purpose: results may outlive static destructors
}  // namespace

void* result_alloc(size_t bytes) {
    const size_t sz = (std::max<size_t>(bytes, 1) + 65535) & ~(size_t)65535;
    PinnedPool& pp = pool();
    {
        std::lock_guard<std::mutex> lock(pp.mu);
        auto it = pp.free_blocks.lower_bound(sz);
        if (it != pp.free_blocks.end() && it->first <= sz * 2) {
            void* p = it->second;
            pp.free_bytes -= it->first;
            pp.free_blocks.erase(it);
            return p;
        }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, sz, hipHostMallocDefault) != hipSuccess || !p) {
        (void)hipGetLastError();
        return malloc(sz);   // still a valid result buffer for the copy path (result_is_pinned() says no: never handed to a kernel)
    }
    std::lock_guard<std::mutex> lock(pp.mu);
    pp.owned[p] = sz;
    return p;
}

bool result_is_pinned(const void* p) {
    PinnedPool& pp = pool();
    std::lock_guard<std::mutex> lock(pp.mu);
    return p && pp.owned.count(const_cast<void*>(p)) != 0;
}

void result_free(void* p) {
    if (!p) return;
    PinnedPool& pp = pool();
    size_t sz = 0;
    {
        std::lock_guard<std::mutex> lock(pp.mu);
        auto it = pp.owned.find(p);
        if (it == pp.owned.end()) { free(p); return; }
        sz = it->second;
        if (pp.free_bytes + sz <= PinnedPool::kKeep) {
            pp.free_blocks.emplace(sz, p);
            pp.free_bytes += sz;
            return;
        }
        pp.owned.erase(it);
    }
    (void)hipHostFree(p);
}

// ------------------------------------------------------------------------------------------------
// GenerateAudio for a batch of independent utterance chunks
// ------------------------------------------------------------------------------------------------
int resolve_max_steps(const ptts_request& r) {  // runtime_native_safetensors.go:61-67, text/prepare.go:38-48
    int ms = r.max_steps;
    if (ms <= 0) ms = r.estimated_max_steps;
    if (ms <= 0) ms = (int)std::ceil(((double)r.n_tokens / 3.0 + 2.0) * 12.5);
    return ms;
}


// the graph of `nsteps` consecutive AR steps whose attention launches cover `ni` load rounds (captured on first use, kept with
// the batch).  Several steps per graph: the gap between two replays (~8 us of idle GPU) is paid once per graph.
static hipGraphExec_t step_graph(Batch& b, int lsd, int ni, int nsteps) {
    Model& m = *b.m;
    if (b.graph_lsd != lsd) {
        for (auto& set : b.graphs) for (auto& row : set) for (hipGraphExec_t& g : row) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
        b.graph_lsd = lsd;
        for (auto& gs : b.graph_steps) gs[0] = gs[1] = 0;
    }
    // two sets (a step with sampling noise reads its noise row: other launches), three columns each: single steps, the short graphs (5 steps:
    // batches that may end by EOS, and the tail of the others), the long ones (25).  A column is re-captured only when ITS step count changes:
    // calls that alternate temperatures, or 25 / 5 / 1 steps, keep their graphs
    const int ns = b.has_noise ? 1 : 0;
    const int col = nsteps <= 1 ? 0 : nsteps <= 5 ? 1 : 2;
    if (col > 0 && b.graph_steps[ns][col - 1] != nsteps) {
        for (auto& row : b.graphs[ns]) if (row[col]) { (void)hipGraphExecDestroy(row[col]); row[col] = nullptr; }
        b.graph_steps[ns][col - 1] = nsteps;
    }
    hipGraphExec_t& slot = b.graphs[ns][ni][col];
    if (slot) return slot;
    hipGraph_t g = nullptr;
    PTTS_HIP(hipStreamBeginCapture(m.stream, hipStreamCaptureModeThreadLocal));
    b.capturing = true;
    b.capture_keys = ni * attn_step_keys_per_round(m.opts.kv == PTTS_KV_BF16);
    const int ld = m.d.ldim;
    const int64_t ls = (int64_t)b.max_steps * ld;
    try {
        // a graph opens its first step with k_step_begin (slots may have changed since the last one); inside, every step's last launch opens the next
        b.opened = false;
        for (int k = 0; k < nsteps; k++) {
            if (!b.opened) step_open(b);
            if (!step_core(b, lsd, b.opened, true, k + 1 < nsteps))
                launch_step_finish(b.st, b.cur_now(), b.eos.as<float>(), ld, b.B, b.latents.as<float>(), ls, m.stream);
        }
        b.opened = false;
    } catch (...) {
        b.capturing = false;
        b.opened = false;
        (void)hipStreamEndCapture(m.stream, &g);
        if (g) (void)hipGraphDestroy(g);
        throw;
    }
    b.capturing = false;
    PTTS_HIP(hipStreamEndCapture(m.stream, &g));
    hipError_t e = hipGraphInstantiate(&slot, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) throw Error(PTTS_ENODEVICE, strfmt("hip: hipGraphInstantiate failed: %s", hipGetErrorString(e)));
    return slot;
}

// nsteps > 1 only with use_graph
void enqueue_step(Batch& b, int lsd, bool use_graph, int nsteps) {
    Model& m = *b.m;
    if (use_graph) {
        const int ni = attn_step_rounds(std::min(b.kv_bound + nsteps, b.cap), m.opts.kv == PTTS_KV_BF16);
        PTTS_HIP(hipGraphLaunch(step_graph(b, lsd, ni, nsteps), m.stream));
        b.kv_bound += nsteps;
        b.opened = false;   // (a graph ends unchained)
        return;
    }
    const int ld = m.d.ldim;
    const int64_t ls = (int64_t)b.max_steps * ld;
    // plain launches: the previous step's last launch normally opened this one (b.opened); whoever changes a slot between two steps
    // (batch_reset, admission of a newcomer, the staged API) clears the flag, and k_step_begin opens the step for every row
    if (!b.opened) step_open(b);
    if (!step_core(b, lsd, b.opened, true, true))   // the bookkeeping rides in the step's last launch (shapes it does not take: k_step_finish)
        launch_step_finish(b.st, b.cur_now(), b.eos.as<float>(), ld, b.B, b.latents.as<float>(), ls, m.stream);
}

static void fail_req(ptts_result& r, int code) {
    r.status = code;
}

static void generate_chunk(Model& m, const ptts_request* reqs, const std::vector<int>& idx, ptts_result* res, int lsd) {
    const Desc& d = m.d;
    hipStream_t s = m.stream;
    UploadScope upload_scope(m.upload, s);
    // PTTS_TRACE=1: host wall time of the phases of one call (stderr); each mark drains the stream first
    static const bool trace = getenv("PTTS_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(s);
        (void)hipStreamSynchronize(m.stream2);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[ptts] %-12s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const int B = (int)idx.size(), D = d.d_model, ld = d.ldim;
    std::vector<int> ms((size_t)B), tp((size_t)B), off((size_t)B);
    int cap_need = 0, ms_max = 0;
    for (int i = 0; i < B; i++) {
        const ptts_request& r = reqs[idx[i]];
        ms[i] = resolve_max_steps(r);
        tp[i] = (int)r.n_tokens + (r.voice_embedding ? (int)r.voice_frames : 0);
        off[i] = r.voice ? reinterpret_cast<const Voice*>(r.voice)->offset : (r.voice_caches ? (int)r.voice_offsets[0] : 0);
        cap_need = std::max(cap_need, off[i] + tp[i] + ms[i]);
        ms_max = std::max(ms_max, ms[i]);
    }
    int cap = (cap_need + 63) / 64 * 64;
    if (cap > ROPE_SEQ) throw Error(PTTS_EINVAL, strfmt("ops: rope cos/sin sequence length too small for pos=%d seq=1", cap_need));
    if (!m.cached_batch || m.cached_batch->B != B || m.cached_batch->cap < cap || m.cached_batch->max_steps < ms_max) {
        m.cached_batch.reset();
        m.cached_batch.reset(batch_new(m, B, cap, ms_max));
    }
    Batch& b = *m.cached_batch;
    batch_reset(b);
    {
        // (a slot stops at most one frame past what the decoder's RoPE table reaches: generating that frame is what the reference reports as the
        // error, mimi.go:498 -- for that utterance alone; the other utterances of the batch are unaffected)
        const int dec_limit = ROPE_SEQ / d.up_stride;
        std::vector<int32_t> v_ms((size_t)B), v_fae((size_t)B);
        for (int i = 0; i < B; i++) v_ms[(size_t)i] = std::min(ms[(size_t)i], dec_limit + 1);
        std::vector<float> v_thr((size_t)B);
        for (int i = 0; i < B; i++) { v_fae[i] = reqs[idx[i]].frames_after_eos; v_thr[i] = reqs[idx[i]].eos_threshold; }
        h2d(b.st.max_steps, v_ms.data(), (size_t)B * 4, s);
        h2d(b.st.frames_after_eos, v_fae.data(), (size_t)B * 4, s);
        h2d(b.st.eos_threshold, v_thr.data(), (size_t)B * 4, s);
    }
    {
        std::map<const Voice*, std::vector<int32_t>> by_voice;
        for (int i = 0; i < B; i++) {
            const ptts_request& r = reqs[idx[i]];
            if (r.voice) by_voice[reinterpret_cast<const Voice*>(r.voice)].push_back(i);
            else if (r.voice_caches) batch_set_voice(b, i, r.voice_caches, r.voice_cache_steps, r.voice_offsets);
        }
        for (auto& kv : by_voice) batch_apply_voice(b, *kv.first, kv.second);
    }
    // text (+ voice) embeddings packed as rows (runtime_native_safetensors.go:89-119)
    std::vector<int64_t> row_off((size_t)B + 1, 0);
    for (int i = 0; i < B; i++) row_off[i + 1] = row_off[i] + tp[i];
    const int64_t R = row_off[B];
    DevBuf& rows = m.work(5, (size_t)R * D * sizeof(float));
    {
        std::vector<int64_t> ids;
        for (int i = 0; i < B; i++) ids.insert(ids.end(), reqs[idx[i]].tokens, reqs[idx[i]].tokens + reqs[idx[i]].n_tokens);
        DevBuf& dids = m.work(6, ids.size() * sizeof(int64_t));
        h2d(dids.p, ids.data(), ids.size() * sizeof(int64_t), s);
        // token rows of consecutive requests are contiguous unless a voice embedding sits between them: one gather per such run
        // (a batch without voice embeddings is ONE launch instead of one per request)
        int64_t id0 = 0, run_id0 = 0, run_n = 0;
        float* run_dst = rows.as<float>();
        auto flush = [&]() {
            if (run_n > 0) launch_embed_gather(m.at<float>(d.embed), dids.as<int64_t>() + run_id0, (int)run_n, D, run_dst, s);
            run_n = 0;
        };
        for (int i = 0; i < B; i++) {
            const ptts_request& r = reqs[idx[i]];
            float* dst = rows.as<float>() + row_off[i] * D;
            int64_t tv = r.voice_embedding ? r.voice_frames : 0;
            if (tv) {
                flush();
                h2d(dst, r.voice_embedding, (size_t)tv * D * sizeof(float), s);
            }
            if (run_n == 0) { run_id0 = id0; run_dst = dst + tv * D; }
            run_n += r.n_tokens;
            id0 += r.n_tokens;
        }
        flush();
    }
    mark("setup");
    auto phase = [&](int i, hipStream_t st) {   // measurement pass only (ptts_profile_enable): device time of the call's phases
        if (!m.prof.phases_on) return;
        if (!m.prof.phase[i]) PTTS_HIP(hipEventCreate(&m.prof.phase[i]));
        PTTS_HIP(hipEventRecord(m.prof.phase[i], st));
    };
    m.prof.phases = false;
    phase(0, s);
    batch_prompt(b, rows.as<float>(), row_off.data());
    phase(1, s);
    mark("prefill");
    // sampling noise (flow_lm.go:283-288,386-408): injected rows as they are; otherwise N(0,1) * sqrt(temperature) drawn on the
    // device per (seed, step); temperature <= 0: zeros.  All of it is resident before the first step, so the AR loop (plain
    // launches or graph replay) just reads row `step` of its slot.
    bool any_noise = false, any_draw = false;
    for (int i = 0; i < B; i++) {
        const ptts_request& r = reqs[idx[i]];
        any_noise |= r.noise != nullptr || r.temperature > 0.0f;
        any_draw |= r.noise == nullptr && r.temperature > 0.0f;
    }
    b.has_noise = any_noise;
    if (any_noise) {
        size_t n = (size_t)B * b.max_steps * ld;
        b.noise.ensure(n * sizeof(float));
        PTTS_HIP(hipMemsetAsync(b.noise.p, 0, n * sizeof(float), s));
        for (int i = 0; i < B; i++)
            if (reqs[idx[i]].noise) h2d(b.noise.as<float>() + (size_t)i * b.max_steps * ld, reqs[idx[i]].noise, (size_t)ms[i] * ld * sizeof(float), s);
        if (any_draw) {
            if (ld % 4) throw Error(PTTS_EINVAL, "ptts-hip: the device noise draw needs a latent width that is a multiple of 4");
            std::vector<NoiseSpec> spec((size_t)B);
            for (int i = 0; i < B; i++) {
                const ptts_request& r = reqs[idx[i]];
                const bool draw = r.noise == nullptr && r.temperature > 0.0f;
                spec[(size_t)i] = NoiseSpec{draw ? (r.noise_seed ? r.noise_seed : m.next_noise_seed()) : 0, draw ? std::sqrt(r.temperature) : 0.0f, draw ? ms[i] : 0};
            }
            DevBuf& sb = m.work(12, spec.size() * sizeof(NoiseSpec));
            h2d(sb.p, spec.data(), spec.size() * sizeof(NoiseSpec), s);
            launch_noise_fill(sb.as<NoiseSpec>(), B, ms_max, b.noise.as<float>(), (int64_t)b.max_steps * ld, ld, s);
        }
    }
    m.tcomb_for(lsd);
    // PTTS_GRAPH=0/1 overrides the option (A/B measurement, tools/eager_vs_graph.py)
    static const int env_graph = [] { const char* e = getenv("PTTS_GRAPH"); return e ? atoi(e) : -1; }();
    const bool use_graph = (env_graph >= 0 ? env_graph != 0 : m.opts.use_graph != 0) && !m.prof.on;
    bool may_stop = false, any_cb = false;
    for (int i = 0; i < B; i++) {
        may_stop |= reqs[idx[i]].eos_threshold < 1e30f;
        any_cb |= reqs[idx[i]].step_callback != nullptr;
    }
    std::vector<char> cancelled((size_t)B, 0);
    std::vector<int32_t> act((size_t)B, 1);
    // Mimi decode of finished frame ranges runs on a second stream while the AR loop keeps stepping: the loop is a chain
    // of latency-bound launches that leaves most of the chip idle, the decoder is throughput work, and every decoder op is
    // causal, so frames [f0, f1) can be decoded as soon as step f1-1 has finished.
    const int64_t spf = d.samples_per_frame;
    // The decoder's RoPE table ends at ROPE_SEQ positions (mimi.go:498): like the reference, a step budget beyond that is not an
    // error by itself -- generating that many frames is (the decoder is sized for what can be decoded).
    const int t_limit = ROPE_SEQ / d.up_stride;
    const int T = std::min(ms_max, t_limit);
    auto too_long = [&](int frames) {
        return Error(PTTS_EINVAL, strfmt("generate: mimi_decode: ops: rope cos/sin sequence length too small for pos=0 seq=%lld", (long long)frames * d.up_stride));
    };
    DevBuf& pcm = m.work(7, (size_t)B * T * spf * sizeof(float));
    const char* env_chunk = getenv("PTTS_MIMI_CHUNK");
    int chunk = env_chunk && atoi(env_chunk) > 0 ? atoi(env_chunk) : 1 << 30;   // default: decode after the loop (measured: overlapping
                                                                                // slows the AR launches by as much as it hides)
    // Streaming (pcm_callback): ranges of `chunk` frames are decoded on stream2 behind the AR loop, copied into the buffers
    // the results will own and announced from a host function queued on that stream (a HIP runtime thread).
    struct StreamChunk {
        const ptts_request* reqs; const int* idx; const char* cancelled; void* const* host; int B;
        int f0, f1; int64_t spf;
        int32_t* nf;   // pinned: n_frames of every slot, read after the range's last step
    };
    bool streaming = false;
    for (int i = 0; i < B; i++) streaming |= reqs[idx[i]].pcm_callback != nullptr;
    std::vector<void*> stream_host((size_t)B, nullptr);
    std::vector<std::unique_ptr<StreamChunk>> stream_chunks;
    std::vector<int32_t*> stream_nf;
    DevBuf* stream_s16 = nullptr;
    // Error path: a throw below may leave copies, decoder launches and host functions queued on the two streams that still read
    // `cancelled`, `stream_chunks`, `stream_host` (declared above, destroyed after this guard): drain both streams first, then
    // give the streaming buffers back.  Dismissed on the normal path, which does the same things in order.
    struct Unwind {
        Model& m; std::vector<void*>& host; std::vector<int32_t*>& nf; bool armed = true;
        ~Unwind() {
            if (!armed) return;
            (void)hipStreamSynchronize(m.stream2);
            (void)hipStreamSynchronize(m.stream);
            for (int32_t* p : nf) (void)hipHostFree(p);
            nf.clear();
            for (void*& p : host) { result_free(p); p = nullptr; }
        }
    } unwind{m, stream_host, stream_nf};
    if (streaming) {
        chunk = 1 << 30;
        bool any_s16 = false;
        for (int i = 0; i < B; i++) {
            const ptts_request& r = reqs[idx[i]];
            if (!r.pcm_callback) continue;
            chunk = std::min(chunk, r.stream_frames > 0 ? (int)r.stream_frames : 12);
            const size_t esz = r.pcm_format == PTTS_PCM_S16 ? sizeof(int16_t) : sizeof(float);
            stream_host[(size_t)i] = result_alloc((size_t)std::max<int64_t>(1, (int64_t)ms[i] * spf) * esz);
            if (!stream_host[(size_t)i]) throw Error(PTTS_ENOMEM, "ptts-hip: out of host memory");
            any_s16 |= r.pcm_format == PTTS_PCM_S16;
        }
        if (any_s16) stream_s16 = &m.work(8, (size_t)B * T * spf * sizeof(int16_t));
    }
    // Decoder workspace (~2.7 MB of f32 activations per latent frame and utterance).  A batch of more than kMimiGroup utterances that is decoded in one go after the
    // loop (the default) goes through the decoder group after group in the SAME buffers (stream order keeps them apart); a batch whose frame ranges are decoded under
    // the loop (streaming, PTTS_MIMI_CHUNK) needs every utterance's history from range to range and keeps one workspace for all of them.
    const int dec_group = (B > kMimiGroup && !streaming && chunk > ms_max) ? kMimiGroup : B;
    MimiWs mw;
    mimi_setup(m, mw, dec_group, T);
    mimi_zero_history(m, mw, m.stream2);   // nine small launches: under the AR loop instead of between the loop and the decoder
    int f_done = 0, f_emitted = 0, steps_run = 0;
    size_t ev_used = 0;
    auto next_event = [&]() {
        if (m.events.size() <= ev_used) { hipEvent_t e; PTTS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); m.events.push_back(e); }
        return m.events[ev_used++];
    };
    const PcmRow* pcm_rows = nullptr;
    bool* rows_used = nullptr;
    auto decode_upto = [&](int f1) {
        if (f1 <= f_done) return;
        if (f1 > T) throw too_long(f1);
        hipEvent_t e = next_event();
        PTTS_HIP(hipEventRecord(e, s));
        PTTS_HIP(hipStreamWaitEvent(m.stream2, e, 0));
        const int64_t lstride = (int64_t)b.max_steps * ld;
        int direct = 0, groups = 0;
        for (int g0 = 0; g0 < B; g0 += dec_group, groups++) {
            const int nb = std::min(dec_group, B - g0);
            MimiWs wg;
            MimiWs* w = &mw;
            if (nb != mw.B) { mimi_setup(m, wg, nb, T); w = &wg; }   // the last, smaller group: its own layout inside the same (grow-only) buffer
            if (g0 > 0) w->zeroed = false;                           // (a group's history rows: zeroed again in front of its decode)
            bool used = false;
            mimi_range(m, *w, b.latents.as<float>() + (int64_t)g0 * lstride, lstride, f_done, f1, pcm.as<float>() + (size_t)g0 * T * spf, nullptr, m.stream2,
                       (f_done == 0 && pcm_rows) ? pcm_rows + g0 : nullptr, &used);
            direct += used ? 1 : 0;
        }
        if (direct != 0 && direct != groups) throw Error(PTTS_EINVAL, "ptts-hip: internal: the decoder's groups disagree about the direct PCM rows");
        if (rows_used) *rows_used = direct != 0;
        f_done = f1;
    };
    auto emit_upto = [&](int f1) {   // hand frames [f_emitted, f1) to the streaming callbacks (they are decoded: f1 <= f_done)
        if (!streaming || f1 <= f_emitted) return;
        const int f0 = f_emitted;
        hipStream_t s2 = m.stream2;
        if (stream_s16) launch_pcm16_rows(pcm.as<float>(), stream_s16->as<int16_t>(), B, (int64_t)T * spf, (int64_t)f0 * spf, (int64_t)(f1 - f0) * spf, s2);
        for (int i = 0; i < B; i++) {
            if (!stream_host[(size_t)i]) continue;
            const int fe = std::min(f1, ms[i]);
            if (fe <= f0) continue;
            const bool s16 = reqs[idx[i]].pcm_format == PTTS_PCM_S16;
            const size_t esz = s16 ? sizeof(int16_t) : sizeof(float);
            const char* src = s16 ? (const char*)stream_s16->p : (const char*)pcm.p;
            PTTS_HIP(hipMemcpyAsync((char*)stream_host[(size_t)i] + (size_t)f0 * spf * esz, src + ((size_t)i * T * spf + (size_t)f0 * spf) * esz,
                                    (size_t)(fe - f0) * spf * esz, hipMemcpyDeviceToHost, s2));
        }
        int32_t* nfp = nullptr;
        PTTS_HIP(hipHostMalloc((void**)&nfp, (size_t)B * sizeof(int32_t), hipHostMallocDefault));
        stream_nf.push_back(nfp);
        PTTS_HIP(hipMemcpyAsync(nfp, b.st.n_frames, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, s2));
        stream_chunks.emplace_back(new StreamChunk{reqs, idx.data(), cancelled.data(), stream_host.data(), B, f0, f1, spf, nfp});
        PTTS_HIP(hipLaunchHostFunc(s2, [](void* p) {
            const StreamChunk& c = *static_cast<const StreamChunk*>(p);
            for (int i = 0; i < c.B; i++) {
                const ptts_request& r = c.reqs[c.idx[i]];
                if (!r.pcm_callback || !c.host[i] || c.cancelled[i]) continue;
                const int end = std::min(c.f1, (int)c.nf[i]);   // frames past the utterance's end are not audio
                if (end <= c.f0) continue;
                const size_t esz = r.pcm_format == PTTS_PCM_S16 ? sizeof(int16_t) : sizeof(float);
                r.pcm_callback(r.pcm_user, (int64_t)c.f0 * c.spf, (int64_t)(end - c.f0) * c.spf, (const char*)c.host[i] + (size_t)c.f0 * c.spf * esz);
            }
        }, stream_chunks.back().get()));
        f_emitted = f1;
    };
    // graph replay without per-step host work (no step callbacks, no cancel flags, no ranges to decode on the way): several
    // steps per graph.  Slots that finish inside a graph are skipped by every kernel of the remaining steps (active flags).
    // 5 steps per graph; 25 when no request of the batch can end by EOS (threshold = +inf: every budget is known, so no replayed step can turn out to
    // have been for nothing) -- measured on the 125-step batch: 1 -> 5 steps: -0.2..-0.6 ms, 5 -> 25: -0.5 ms, 125: -0.15 (one big graph is slower again)
    static const int env_gsteps = [] { const char* e = getenv("PTTS_GRAPH_STEPS"); return e ? std::max(1, atoi(e)) : 0; }();
    bool any_cancel_flag = false;
    for (int i = 0; i < B; i++) any_cancel_flag |= reqs[idx[i]].cancel != nullptr;
    const int gsteps = (use_graph && !any_cb && !any_cancel_flag && chunk > ms_max) ? (env_gsteps ? env_gsteps : (may_stop ? 5 : 25)) : 1;
    const int ms_loop = std::min(ms_max, t_limit + 1);
    for (int step = 0; step < ms_loop;) {
        int n_cancel = 0;
        for (int i = 0; i < B; i++) {  // ctx.Err() check before every step (:156-159)
            const ptts_request& r = reqs[idx[i]];
            if (r.cancel && *r.cancel) cancelled[i] = 1;
            n_cancel += cancelled[i];
        }
        if (n_cancel == B) break;
        const int n_now = (gsteps > 1 && step + gsteps <= ms_loop) ? gsteps : (gsteps > 5 && step + 5 <= ms_loop) ? 5 : 1;   // the tail: smaller graphs
        enqueue_step(b, lsd, use_graph, n_now);
        step += n_now;
        steps_run = step;
        if (steps_run % chunk == 0) { decode_upto(steps_run); emit_upto(steps_run); }
        if (any_cb) {  // StepCallback runs synchronously after the step (:194-196)
            std::vector<int32_t> before = act, broke((size_t)B);
            d2h(act.data(), b.st.active, (size_t)B * 4, s);
            d2h(broke.data(), b.st.broke, (size_t)B * 4, s);
            for (int i = 0; i < B; i++) {  // not called for the iteration that leaves through `break` (:185-187)
                const ptts_request& r = reqs[idx[i]];
                if (r.step_callback && before[i] && !broke[i] && !cancelled[i]) r.step_callback(r.callback_user, step, ms[i]);
            }
            bool any = false;
            for (int i = 0; i < B; i++) any |= act[i] != 0;
            if (!any) break;
        } else if (may_stop && step / 8 != (step - n_now) / 8) {   // every eighth step (or the first graph boundary past it)
            PTTS_HIP(hipMemcpyAsync(b.n_active_pinned, b.st.n_active, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            PTTS_HIP(hipStreamSynchronize(s));
            if (*b.n_active_pinned <= 0) break;
        }
    }
    phase(2, s);
    mark("ar loop");
    // n_frames and eos_step sit side by side in the state block: one copy into page-locked memory, one wait
    PTTS_HIP(hipMemcpyAsync(b.n_active_pinned + 1, b.st.n_frames, (size_t)2 * B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    b.n_active_pinned[1 + 2 * B] = 0;
    if (b.fc_ok) PTTS_HIP(hipMemcpyAsync(b.n_active_pinned + 1 + 2 * B, b.fc_fault(), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    PTTS_HIP(hipStreamSynchronize(s));
    if (b.n_active_pinned[1 + 2 * B]) flow_cluster_fault(b);
    const std::vector<int32_t> nf(b.n_active_pinned + 1, b.n_active_pinned + 1 + B), es(b.n_active_pinned + 1 + B, b.n_active_pinned + 1 + 2 * B);
    int Tmax = 0;
    std::vector<char> overlong((size_t)B, 0);   // utterances that ran past the decoder's reach: failed one by one, like the reference's one GenerateAudio call
    for (int i = 0; i < B; i++) {
        overlong[(size_t)i] = !cancelled[i] && nf[i] > t_limit;
        if (overlong[(size_t)i]) set_last_error(too_long(nf[i]).what());
        if (!cancelled[i] && !overlong[(size_t)i]) Tmax = std::max(Tmax, nf[i]);
    }
    if (Tmax > 0) {
        DevBuf* pcm_s16 = nullptr;   // PCM16 egress on the device (audio/wav_stream.go:43-54) for the requests that ask for it
        bool any_s16 = false;
        for (int i = 0; i < B; i++) any_s16 |= reqs[idx[i]].pcm_format == PTTS_PCM_S16 && !cancelled[i] && !stream_host[(size_t)i];
        if (any_s16) pcm_s16 = &m.work(8, (size_t)B * T * spf * sizeof(int16_t));
        std::vector<void*> host_dst((size_t)B, nullptr);   // page-locked result buffers allocated ahead of the decoder (direct rows)
        bool direct_done = false;                          // the decoder's last kernel wrote the samples into host_dst itself
        // results of utterances [b0, b1): buffers, and the copies queued on s
        auto finish_rows = [&](int b0, int b1) {
        for (int i = b0; i < b1; i++) {
            ptts_result& r = res[idx[i]];
            if (cancelled[i]) { fail_req(r, PTTS_ECANCELLED); continue; }
            if (overlong[(size_t)i]) { fail_req(r, PTTS_EINVAL); result_free(host_dst[(size_t)i]); host_dst[(size_t)i] = nullptr; continue; }
            r.n_frames = nf[i];
            r.eos_step = es[i];
            r.n_samples = (int64_t)nf[i] * spf;
            if (stream_host[(size_t)i]) {   // streamed: the buffer already holds every sample that was announced
                if (reqs[idx[i]].pcm_format == PTTS_PCM_S16) r.pcm16 = (int16_t*)stream_host[(size_t)i];
                else r.pcm = (float*)stream_host[(size_t)i];
                stream_host[(size_t)i] = nullptr;
            } else if (reqs[idx[i]].pcm_format == PTTS_PCM_S16) {
                r.pcm16 = (int16_t*)(host_dst[(size_t)i] ? host_dst[(size_t)i] : result_alloc((size_t)std::max<int64_t>(1, r.n_samples) * sizeof(int16_t)));
                host_dst[(size_t)i] = nullptr;
                if (!r.pcm16) { fail_req(r, PTTS_ENOMEM); continue; }
                if (r.n_samples > 0 && !direct_done)
                    PTTS_HIP(hipMemcpyAsync(r.pcm16, pcm_s16->as<int16_t>() + (size_t)i * T * spf, (size_t)r.n_samples * sizeof(int16_t), hipMemcpyDeviceToHost, s));
            } else {
                r.pcm = (float*)(host_dst[(size_t)i] ? host_dst[(size_t)i] : result_alloc((size_t)std::max<int64_t>(1, r.n_samples) * sizeof(float)));
                host_dst[(size_t)i] = nullptr;
                if (!r.pcm) { fail_req(r, PTTS_ENOMEM); continue; }
                if (r.n_samples > 0 && !direct_done)   // all copies are queued back to back; one wait below
                    PTTS_HIP(hipMemcpyAsync(r.pcm, pcm.as<float>() + (size_t)i * T * spf, (size_t)r.n_samples * sizeof(float), hipMemcpyDeviceToHost, s));
            }
            if (reqs[idx[i]].want_latents) {
                r.latents = (float*)malloc((size_t)std::max(1, nf[i]) * ld * sizeof(float));
                if (!r.latents) { fail_req(r, PTTS_ENOMEM); continue; }
                d2h(r.latents, b.latents.as<float>() + (size_t)i * b.max_steps * ld, (size_t)nf[i] * ld * sizeof(float), s);
            }
            r.status = PTTS_OK;
        }
        };
        // Whole batch decoded in one go (the default): the decoder's last kernel stores every utterance's samples straight into
        // its page-locked result buffer (f32 or int16) -- the kernel's stores ARE the device->host transfer: no device PCM buffer,
        // no copies, no copy kernels competing with the decoder.
        const bool try_direct = !streaming && f_done == 0;
        const PcmRow* d_rows = nullptr;
        if (try_direct) {
            PcmRow* rows = b.rows_pinned;
            for (int i = 0; i < B; i++) rows[i] = PcmRow{nullptr, 0, 0};
            for (int i = 0; i < B; i++) {
                if (cancelled[i] || overlong[(size_t)i]) continue;
                const bool s16 = reqs[idx[i]].pcm_format == PTTS_PCM_S16;
                const int64_t ns = (int64_t)nf[i] * spf;
                host_dst[(size_t)i] = result_alloc((size_t)std::max<int64_t>(1, ns) * (s16 ? sizeof(int16_t) : sizeof(float)));
                if (!host_dst[(size_t)i]) continue;   // reported as PTTS_ENOMEM by finish_rows
                rows[i] = PcmRow{host_dst[(size_t)i], (int32_t)std::min<int64_t>(ns, INT32_MAX), s16 ? 1 : 0};
            }
            // a kernel may only store into page-locked memory: if the pool had to fall back to pageable blocks (hipHostMalloc
            // failed), the whole batch takes the device buffer + copy path instead
            bool all_pinned = true;
            for (int i = 0; i < B; i++) all_pinned &= !host_dst[(size_t)i] || result_is_pinned(host_dst[(size_t)i]);
            if (all_pinned) {
                DevBuf& rb = m.work(11, (size_t)B * sizeof(PcmRow));
                PTTS_HIP(hipMemcpyAsync(rb.p, rows, (size_t)B * sizeof(PcmRow), hipMemcpyHostToDevice, s));   // page-locked source: no wait needed
                d_rows = rb.as<PcmRow>();
            }
        }
        pcm_rows = d_rows; rows_used = &direct_done;
        if (m.prof.phases_on) { PTTS_HIP(hipStreamWaitEvent(m.stream2, m.prof.phase[2], 0)); phase(3, m.stream2); }
        decode_upto(std::min(steps_run, Tmax));   // frames past every utterance's end are never decoded
        pcm_rows = nullptr; rows_used = nullptr;
        emit_upto(std::min(steps_run, Tmax));
        phase(4, m.stream2);
        m.prof.phases = m.prof.phases_on && f_done > 0;
        PTTS_HIP(hipStreamSynchronize(m.stream2));
        for (int32_t* p : stream_nf) (void)hipHostFree(p);
        stream_nf.clear();
        mark("mimi");
        if (pcm_s16 && !direct_done) launch_pcm16(pcm.as<float>(), pcm_s16->as<int16_t>(), (int64_t)B * T * spf, s);
        finish_rows(0, B);
        for (void* p : host_dst) result_free(p);   // rows that were allocated but not handed out (failed requests)
        PTTS_HIP(hipStreamSynchronize(s));
        mark("results d2h");
    } else {
        PTTS_HIP(hipStreamSynchronize(m.stream2));
        for (int32_t* p : stream_nf) (void)hipHostFree(p);
        for (int i = 0; i < B; i++) fail_req(res[idx[i]], overlong[(size_t)i] ? PTTS_EINVAL : PTTS_ECANCELLED);
    }
    for (void* p : stream_host) result_free(p);   // buffers of cancelled / failed streaming requests
    unwind.armed = false;
}

std::string request_error(const Desc& d, const ptts_request& q) {   // the argument checks of GenerateAudio (runtime_native_safetensors.go:52-119)
    if (!q.tokens || q.n_tokens <= 0) return "generate: token slice must not be empty";
    if ((q.voice_embedding != nullptr) + (q.voice_caches != nullptr) + (q.voice != nullptr) > 1) return "generate: voice embedding and voice model state are mutually exclusive";
    if (q.voice_caches && (!q.voice_cache_steps || !q.voice_offsets)) return "generate: load voice model state: missing cache steps/offsets";
    if (q.noise && q.noise_rows > 0 && q.noise_rows < resolve_max_steps(q))
        return strfmt("generate: injected noise has %d rows, the step budget is %d", q.noise_rows, resolve_max_steps(q));
    if (std::isnan(q.temperature)) return "generate: temperature is NaN";
    for (int64_t t = 0; t < q.n_tokens; t++)
        if (q.tokens[t] < 0 || q.tokens[t] >= d.n_bins)
            return strfmt("generate: text embeddings: native: token id %lld (%lld) out of range [0,%d)", (long long)t, (long long)q.tokens[t], d.n_bins);
    return std::string();
}

void generate(Model& m, const ptts_request* reqs, int n, ptts_result* res) {
    std::lock_guard<std::mutex> lock(m.mu);
    m.use_device();
    const Desc& d = m.d;
    std::map<int, std::vector<int>> groups;  // lsd_steps -> request indices
    std::string first_err;
    for (int i = 0; i < n; i++) {
        ptts_result& r = res[i];
        std::memset(&r, 0, sizeof r);
        r.eos_step = -1;
        const ptts_request& q = reqs[i];
        std::string err = request_error(d, q);
        if (!err.empty()) {
            r.status = PTTS_EINVAL;
            if (first_err.empty()) first_err = err;
            continue;
        }
        groups[q.lsd_steps <= 0 ? 1 : q.lsd_steps].push_back(i);
    }
    const int mb = std::max(1, m.opts.max_batch);
    for (auto& kv : groups) {
        const std::vector<int>& all = kv.second;
        for (size_t o = 0; o < all.size(); o += (size_t)mb) {
            std::vector<int> idx(all.begin() + (long)o, all.begin() + (long)std::min(all.size(), o + (size_t)mb));
            try {
                try {
                    generate_chunk(m, reqs, idx, res, kv.first);
                } catch (const FlowClusterFault&) {
                    // a hand-off inside k_flow_cluster timed out somewhere in this chunk's AR loop (found when the loop's counters were read back).  The chunk's
                    // inputs are the requests themselves: run it again on the 2 x depth launches, which compute the same bits (the batch has been switched
                    // over: flow_cluster_recover).  Only a chunk that has already spoken to its caller -- step callbacks, streamed samples -- cannot be re-run.
                    bool spoke = false;
                    for (int i : idx) spoke |= reqs[i].step_callback != nullptr || reqs[i].pcm_callback != nullptr;
                    if (spoke) throw;
                    for (int i : idx) { ptts_free_result(&res[i]); std::memset(&res[i], 0, sizeof res[i]); res[i].eos_step = -1; }
                    generate_chunk(m, reqs, idx, res, kv.first);
                }
            } catch (const Error& e) {
                for (int i : idx) { ptts_free_result(&res[i]); res[i].status = e.code; res[i].eos_step = -1; }
                if (first_err.empty()) first_err = e.what();
            }
        }
    }
    if (!first_err.empty()) set_last_error(first_err);
}

}  // namespace ptts
