// continuous.cpp -- continuous batching for the serving path.
//
// The reference admits `workers` Synthesize calls at a time (internal/server/server.go:398-421) and runs each call's chunks one after
// the other (internal/tts/service.go:138-153); utterances end where their EOS logit says (runtime_native_safetensors.go:178-190), so
// their lengths differ.  The batch-at-a-time dispatcher (dispatcher.cpp) forms a batch, steps it until its LAST utterance has ended
// and only then forms the next: finished slots idle for the rest of the batch, waiting callers wait for all of it.  Here one long-lived
// batch of `slots` utterance states is stepped for as long as there is work:
//   * between groups of AR steps the slots' counters are read back (one 1-KB copy); a slot whose utterance has ended hands its latent
//     frames to the Mimi decoder (second stream, beside the following steps) and is free again;
//   * waiting requests move into free slots at once: their prompts are prefilled as one ragged launch sequence over the new slots only
//     (batch_prompt with empty segments for the running ones), their bookkeeping is (re)initialised by one small kernel;
//   * every caller's audio equals what the same request returns on its own (same kernels, same per-slot arithmetic: rows of a batch
//     never mix) -- tests/test_gpu_continuous.py.
// Nothing in the loop waits for the GPU to run dry: the counters of group g are read (asynchronously, into page-locked memory) while
// group g + 1 is already queued, admissions and decodes are queued behind the group in flight, per-slot device state is written slot
// by slot by one small kernel (never as whole arrays: the other slots are live), and small uploads go through two page-locked arenas
// used in turn.  Prefills and decodes compete with the step chain for the chip.  What round 4's group trace (PTTS_CONT_TRACE, tools/cont_trace_summary.py)
// showed and changed: newcomers are admitted every turn and start stepping with the very next group (the step STREAM waits for their prefill, not the
// host); finished utterances are decoded sixteen at a time, in sub-groups of like length (a decode pads to its longest member), on a stream of the
// engine's own that is confined to half of the CUs; and the decoder's input is gathered on the step stream, which no longer waits for anything the
// decoder does.
// The engine has a fixed geometry (slots, KV capacity, step budget); requests that do not fit it or need per-step host work (step /
// PCM callbacks, lsd_steps > 1) are left to the batch-at-a-time path.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <deque>
#include <map>
#include <mutex>

#include "runtime.h"

namespace ptts {

namespace {
// The pacing below was swept on one MI355X at 128 clients, utterances of 2-12 s (profiles/r4_serve_sweep.txt; tools/gpu_cont_trace.sh), and again in round 5 at
// 128-256 slots (profiles/r5_serve_sweep.txt: decodes of 16 / 32 / 48 utterances alike; 85 / 96 CUs for the decoder -12..-30 %, 160 / 256 CUs -3..-5 %;
// every decode in the step chain with 32 / 48 / 64 / 96 utterances per decode: 13.5 / 14.1 / 14.2 k x at 192 slots against 14.3 k, 14.9 / 15.2 k at 256 against 14.4 k):
// (the decoder's CUs once more on the final engine of round 5, mixed traffic: 192 slots 96 / 128 / 160 / 176 / 192 / 208 / 224 / 256 CUs = 13.0 / 15.7 / 16.0 / 16.0 / 16.5 / 16.3 /
// 15.6 / 15.4 k x; 256 slots 128 / 160 / 192 / 224 = 15.5 / 16.4 / 16.3 / 16.7 k -- but 192 CUs cost uniform traffic 5 % (19.4 k against 20.5 k) and smaller engines 5-10 %
// (128 slots 13.3 k against 14.8 k, 64 slots 10.4 k against 11.0 k): half of the CUs stays)
constexpr int kDecoderShare = 2;      // the decoder's stream gets 1 / kDecoderShare of the device's CUs (128 of the MI355X's 256): see cont_create
constexpr int kDecodeMin = 16;        // finished utterances worth a decode ...
constexpr int kDecodeSerialFrames = 2400;   // finished utterances with this many frames between them (a cluster of long, like-length requests ending together) are decoded IN the step chain, on the whole chip: see start_decode
constexpr int kDecodeMaxAge = 8;      // ... or the oldest has waited this many groups of steps (a caller waiting for audio is a caller not sending its next request)
// a decode pads everything to its longest member: members within max(16 frames, 48 %) of it share one.  Round 4 chose 16 % by the padded frame count (64 slots);
// by TIME the small sub-decodes that rule makes (~370 frames each: 4.6 per decode start at 192 slots) cost more than the padding they save -- the decoder's stream was
// the engine's bottleneck on mixed traffic.  192 slots, mixed 2-12 s: 16 % 14.5 k x, 33 % 14.9 k, 45 % 15.4 k, 60 % 15.1-15.3 k, 75 % 14.5 k, 100 % 14.3 k.
constexpr int kBandFrames = 16, kBandPercent = 48;

// the decoder's CU-masked stream of a GPU, shared by the continuous engines on it (created with the first, destroyed with the last)
std::mutex g_dec_mu;
std::map<int, std::pair<hipStream_t, int>> g_dec_streams;
hipStream_t decoder_stream_acquire(int device) {
    std::lock_guard<std::mutex> lock(g_dec_mu);
    auto& ent = g_dec_streams[device];
    if (!ent.first) {
        // the mask follows the device: its CU count from the properties (bit i = CU i in the runtime's enumeration, which deals CUs round the XCDs, so the
        // low half of the bits is half of every XCD), and the stream is created on THAT device whatever the calling thread's current one is
        int cus = 0, cur = 0;
        PTTS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        if (cus <= 0) throw Error(PTTS_ENODEVICE, "ptts-hip: device reports no compute units");
        PTTS_HIP(hipGetDevice(&cur));
        if (cur != device) PTTS_HIP(hipSetDevice(device));
        std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
        const int open = std::max(1, cus / kDecoderShare);
        for (int i = 0; i < open; i++) mask[(size_t)i >> 5] |= 1u << (i & 31);
        const hipError_t e = hipExtStreamCreateWithCUMask(&ent.first, (uint32_t)mask.size(), mask.data());
        if (cur != device) (void)hipSetDevice(cur);
        if (e != hipSuccess) { ent.first = nullptr; g_dec_streams.erase(device); throw Error(PTTS_ENODEVICE, strfmt("hip: hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e))); }
    }
    ent.second++;
    return ent.first;
}
void decoder_stream_release(int device) {
    std::lock_guard<std::mutex> lock(g_dec_mu);
    auto it = g_dec_streams.find(device);
    if (it == g_dec_streams.end()) return;
    if (--it->second.second <= 0) { (void)hipStreamDestroy(it->second.first); g_dec_streams.erase(it); }
}
constexpr int kFramesPerDecode = 3072;   // frames (utterances x longest) one Mimi decode of finished slots may take: bounds its workspace (~2.7 MB of activations per frame)
}

struct ContEngine {
    Model& m;
    std::unique_ptr<Batch> b;
    int B = 0, cap = 0, max_steps = 0;
    struct Slot {
        bool busy = false;        // holds an utterance that is still generating
        const ptts_request* req = nullptr;
        ptts_result* res = nullptr;
        void* tag = nullptr;
        int ms = 0;               // step budget
        int base_kv = 0;          // cache length after the prefill: the slot's device kv_len is base_kv + n_frames
        uint64_t admit_seq = 0;   // groups launched before the slot was filled: read-backs of those groups predate it
        bool finished = false;    // (transient, inside one read-back) ended: its frames are copied out and the slot is free again
        int nf = 0, eos = -1;     // ... with this many frames / this EOS step
        uint64_t fin_seq = 0;
        int nf_seen = 0; uint64_t seen_seq = 0;   // last read-back that covered the slot: frames then, group then
        bool joining = false;     // its prompt is still being prefilled on the I/O stream: not stepping yet
    };
    std::vector<Slot> slots;
    int n_gen = 0;
    struct Pending {              // a group of finished utterances whose audio is being decoded on the second stream
        hipEvent_t done = nullptr;
        std::vector<void*> tags;
        std::vector<ptts_result*> res;
    };
    std::deque<Pending> decoding;
    // ended utterances whose frames wait (in a staging row of their own, so that the slot is free at once) for a decode worth launching
    struct Staged { const ptts_request* req; ptts_result* res; void* tag; int nf, eos, row; uint64_t seq; };
    std::vector<Staged> staged;
    DevBuf stage;                         // [2 B rows][max_steps][ldim]
    std::vector<int> stage_free;
    // newcomers whose voice ingestion and prefill are running on the I/O stream; they start stepping once that is done
    struct Joining { hipEvent_t ready = nullptr; std::vector<int> slots; int ring = 0; uint64_t seq = 0; };
    std::deque<Joining> joining;
    hipStream_t io = nullptr;             // prefills of newcomers: the model's second stream
    hipStream_t dec = nullptr;            // the decoder's stream of this engine: confined to half of the CUs (cont_create)
    DevBuf adm_dev[4];
    int adm_turn = 0;
    std::vector<hipEvent_t> free_events;
    // read-backs: [5 B] active | step | countdown | n_frames | eos_step as they sit in the state block, two in rotation
    struct Snap { int32_t* host = nullptr; hipEvent_t ready = nullptr; uint64_t seq = 0; bool pending = false; };
    Snap snaps[2];
    uint64_t seq = 0;                     // groups of steps launched so far
    int group_steps = 1;
    uint64_t last_admit_seq = 0;
    UploadArena arenas[2];
    hipEvent_t arena_free[2] = {nullptr, nullptr};
    int arena_turn = 0;
    hipEvent_t ev_steps = nullptr;
    static constexpr int kLat = 4;
    DevBuf lat[kLat];                     // the frames of the utterances being decoded, gathered from their staging rows: four decodes' worth, in turn
    hipEvent_t lat_free[kLat] = {nullptr, nullptr, nullptr, nullptr};
    bool lat_busy[kLat] = {false, false, false, false};
    int lat_turn = 0;
    // where the decoder's last kernel stores each utterance's samples: page-locked result rows (PcmRow table: kLat x 2 B entries, host side page-locked, uploaded per sub-decode)
    PcmRow* rows_host = nullptr;
    DevBuf rows_dev;
    std::unique_lock<std::mutex> hold;    // the model's mutex, held while the engine has work in flight
    bool use_graph = false;
    int64_t admissions = 0, admitted = 0;
    // (measurement: PTTS_CONT_TRACE=<file> -- one line per group of steps: gap in front of it and its duration on the step stream, what ran beside it)
    struct GroupRec { hipEvent_t t0, t1; int n_gen, steps, admitted, decodes_started, decoding, joining; float host_turn_us, host_admit_us, host_decode_us, host_wait_us; };
    std::chrono::steady_clock::time_point tr_last_enqueue = std::chrono::steady_clock::now();
    double tr_acc_admit_us = 0, tr_acc_decode_us = 0, tr_acc_wait_us = 0;   // host time since the previous group was queued: in cont_admit, in start_decode, waiting for a read-back
    std::vector<GroupRec> trace;
    const char* trace_path = getenv("PTTS_CONT_TRACE");
    int tr_admitted = 0, tr_decodes = 0;
    int64_t tr_frames = 0, tr_padded = 0, tr_subs = 0;   // decoded frames, frames incl. padding to the sub-group's longest, sub-groups
    double tr_host_gather_us = 0, tr_host_launch_us = 0; int64_t tr_starts = 0;   // host time of start_decode: the gather on the step stream, the decoder's launches
    int64_t steps_run = 0, slot_steps = 0;   // AR steps launched; utterances stepping in them, summed (their ratio: mean occupancy)

    explicit ContEngine(Model& model) : m(model) {}
    ~ContEngine() {
        (void)hipStreamSynchronize(m.stream);
        (void)hipStreamSynchronize(m.stream2);
        if (io) (void)hipStreamSynchronize(io);   // (the model's second stream: not ours to destroy)
        if (dec) { (void)hipStreamSynchronize(dec); decoder_stream_release(m.device); }
        if (trace_path && !trace.empty()) {
            if (FILE* f = fopen(trace_path, "a")) {
                fprintf(f, "# gap_us dur_us n_gen steps admitted decodes_started decoding joining host_us_since_previous_group of_it_in_cont_admit in_start_decode waiting_for_read_backs\n");
                for (size_t i = 0; i < trace.size(); i++) {
                    float gap = 0.f, dur = 0.f;
                    if (i > 0) (void)hipEventElapsedTime(&gap, trace[i - 1].t1, trace[i].t0);
                    (void)hipEventElapsedTime(&dur, trace[i].t0, trace[i].t1);
                    fprintf(f, "%.1f %.1f %d %d %d %d %d %d %.0f %.0f %.0f %.0f\n", 1e3 * gap, 1e3 * dur, trace[i].n_gen, trace[i].steps, trace[i].admitted, trace[i].decodes_started, trace[i].decoding, trace[i].joining,
                            trace[i].host_turn_us, trace[i].host_admit_us, trace[i].host_decode_us, trace[i].host_wait_us);
                }
                fprintf(f, "# decoded frames %lld, with padding %lld, in %lld decodes\n", (long long)tr_frames, (long long)tr_padded, (long long)tr_subs);
                fprintf(f, "# start_decode calls %lld: host time per call %.0f us for the gather + %.0f us for the decoder's launches and copies\n", (long long)tr_starts, tr_host_gather_us / std::max<int64_t>(1, tr_starts), tr_host_launch_us / std::max<int64_t>(1, tr_starts));
                fclose(f);
            }
            for (auto& g : trace) { (void)hipEventDestroy(g.t0); (void)hipEventDestroy(g.t1); }
        }
        for (auto& j : joining) if (j.ready) (void)hipEventDestroy(j.ready);
        for (auto& p : decoding) if (p.done) (void)hipEventDestroy(p.done);
        for (hipEvent_t e : free_events) (void)hipEventDestroy(e);
        if (ev_steps) (void)hipEventDestroy(ev_steps);
        for (hipEvent_t ev : lat_free) if (ev) (void)hipEventDestroy(ev);
        if (rows_host) (void)hipHostFree(rows_host);
        for (Snap& sn : snaps) { if (sn.host) (void)hipHostFree(sn.host); if (sn.ready) (void)hipEventDestroy(sn.ready); }
        for (int i = 0; i < 2; i++) { if (arenas[i].base) (void)hipHostFree(arenas[i].base); if (arena_free[i]) (void)hipEventDestroy(arena_free[i]); }
        b.reset();
        if (hold.owns_lock()) hold.unlock();
    }
    int free_slots() const { int n = 0; for (const Slot& s : slots) n += !s.busy; return n; }
    int n_finished() const { return (int)staged.size(); }
    int busy() const { return n_gen + n_finished() + (int)decoding.size() + (int)joining.size() + (snaps[0].pending || snaps[1].pending ? 1 : 0); }
    hipEvent_t event() {
        if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
        hipEvent_t e;
        PTTS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return e;
    }
    void lock_model() { if (!hold.owns_lock()) { hold = std::unique_lock<std::mutex>(m.mu); m.use_device(); } }
    void unlock_if_idle() { if (busy() == 0 && hold.owns_lock()) hold.unlock(); }
};

ContEngine* cont_create(Model& m, int slots, int kv_cap, int max_steps) {
    if (slots <= 0 || slots > kStepMaxRows) throw Error(PTTS_EINVAL, strfmt("continuous batch: 1..%d slots (the rows one AR step takes)", kStepMaxRows));
    kv_cap = (std::max(kv_cap, 64) + 63) / 64 * 64;
    std::unique_ptr<ContEngine> e(new ContEngine(m));
    std::lock_guard<std::mutex> lock(m.mu);
    m.use_device();
    e->B = slots; e->cap = kv_cap; e->max_steps = std::max(1, max_steps);
    e->b.reset(batch_new(m, slots, kv_cap, e->max_steps));
    Batch& b = *e->b;
    // every slot starts free: nothing is live until a request moves in
    launch_fill_i32(b.st.active, 0, slots, m.stream);
    launch_fill_i32(b.st.n_active, 0, 1, m.stream);
    // the noise rows exist from the start (zeros for greedy requests): a captured step reads them whatever the slot's temperature
    const size_t nz = (size_t)slots * b.max_steps * m.d.ldim * sizeof(float);
    b.noise.ensure(nz);
    PTTS_HIP(hipMemsetAsync(b.noise.p, 0, nz, m.stream));
    b.has_noise = true;
    e->slots.assign((size_t)slots, ContEngine::Slot{});
    for (ContEngine::Snap& sn : e->snaps) {
        PTTS_HIP(hipHostMalloc((void**)&sn.host, sizeof(int32_t) * (5 * (size_t)slots + 1), hipHostMallocDefault));   // + the flow-cluster fault word
        sn.host[5 * (size_t)slots] = 0;
        PTTS_HIP(hipEventCreateWithFlags(&sn.ready, hipEventDisableTiming));
    }
    for (int i = 0; i < 2; i++) PTTS_HIP(hipEventCreateWithFlags(&e->arena_free[i], hipEventDisableTiming));
    b.slot_local = true;
    // Three streams and no more: the step chain's (the model's first), the prefills' (the model's second: idle while the engine holds the model) and the
    // decoder's below.  With a stream of its own for the prefills as well (round 3's layout plus the decoder's stream: four user streams) every step ran as
    // if confined to the decoder's CUs and queued behind its kernels -- 6.1 k x real time instead of 10.9 k, "alone" groups 410 us a step instead of 290
    // (profiles/r4_serve_sweep.txt, "decoder stream modes"; GPU_MAX_HW_QUEUES = 8 did not change it: how this runtime maps streams onto hardware queues
    // once a CU-masked one exists is not established, so the engine stays at the stream count that measures well).
    e->io = m.stream2;
    {
        // Decodes run BESIDE the step chain here (in a one-shot batch they follow it), and a decoder block holds its CU for a fraction of a millisecond
        // while a step launch lasts microseconds: with the whole chip open to the decoder every step launch of that time queued behind decoder blocks
        // (steps 6x slower for the length of a decode, PTTS_CONT_TRACE).  Confined to half of the CUs (CU-mask bits are dealt round the XCDs: 16 of each
        // XCD's 32) the decoder takes twice as long and the steps beside it 1.4x instead of 6x: +5 % throughput at 128 clients (96 / 160 / 192 CUs:
        // +3 / +4 / +0 %; 64: the decoder falls behind), profiles/r4_serve_sweep.txt.
        e->dec = decoder_stream_acquire(m.device);   // ONE per GPU: the engines of a GPU (ptts_model_share) queue their decodes on the same confined stream
    }
    b.io_stream = e->io;
    for (DevBuf& db : e->adm_dev) db.ensure((size_t)slots * sizeof(SlotAdmit));
    e->stage.ensure((size_t)2 * slots * b.max_steps * m.d.ldim * sizeof(float));
    for (int i = 2 * slots - 1; i >= 0; i--) e->stage_free.push_back(i);
    PTTS_HIP(hipEventCreateWithFlags(&e->ev_steps, hipEventDisableTiming));
    for (hipEvent_t& ev : e->lat_free) PTTS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (DevBuf& l : e->lat) l.ensure((size_t)2 * slots * (size_t)max_steps * m.d.ldim * sizeof(float));   // every staging row at full length
    PTTS_HIP(hipHostMalloc((void**)&e->rows_host, sizeof(PcmRow) * ContEngine::kLat * 2 * (size_t)slots, hipHostMallocDefault));
    e->rows_dev.ensure(sizeof(PcmRow) * ContEngine::kLat * 2 * (size_t)slots);
    m.tcomb_for(1);
    // the decoder's workspaces at their largest, so that a later, bigger group of finished utterances never reallocates a buffer an
    // earlier group's kernels are still using
    { MimiWs w; mimi_setup(m, w, 1, kFramesPerDecode + kFramesPerDecode / 8); }
    m.work(7, (size_t)(kFramesPerDecode + kFramesPerDecode / 8) * m.d.samples_per_frame * sizeof(float));
    m.work(8, (size_t)(kFramesPerDecode + kFramesPerDecode / 8) * m.d.samples_per_frame * sizeof(int16_t));
    m.work(13, (size_t)(kFramesPerDecode + kFramesPerDecode / 8) * m.d.ldim * sizeof(float));
    static const int env_graph = [] { const char* g = getenv("PTTS_GRAPH"); return g ? atoi(g) : -1; }();
    e->use_graph = env_graph >= 0 ? env_graph != 0 : m.opts.use_graph != 0;
    PTTS_HIP(hipStreamSynchronize(m.stream));
    return e.release();
}

void cont_destroy(ContEngine* e) { delete e; }
int cont_free_slots(const ContEngine& e) { return e.free_slots(); }
int cont_busy(const ContEngine& e) { return e.busy(); }
void cont_counts(const ContEngine& e, int64_t* admissions, int64_t* admitted) { *admissions = e.admissions; *admitted = e.admitted; }
void cont_occupancy(const ContEngine& e, int64_t* steps, int64_t* slot_steps) { *steps = e.steps_run; *slot_steps = e.slot_steps; }

bool cont_accepts(const ContEngine& e, const ptts_request& r) {
    if (r.step_callback || r.pcm_callback || r.lsd_steps > 1) return false;
    const int ms = resolve_max_steps(r);
    const int tp = (int)r.n_tokens + (r.voice_embedding ? (int)r.voice_frames : 0);
    const int off = r.voice ? reinterpret_cast<const Voice*>(r.voice)->offset : (r.voice_caches ? (int)r.voice_offsets[0] : 0);
    return ms <= e.max_steps && off + tp + ms <= e.cap && ms <= ROPE_SEQ / e.m.d.up_stride;
}

// n waiting requests move into free slots (n <= cont_free_slots): voice state, prompt prefill, bookkeeping, noise rows
void cont_admit(ContEngine& e, const ptts_request* const* reqs, ptts_result* const* results, void* const* tags, int n) {
    struct Timer { ContEngine& e; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                   ~Timer() { e.tr_acc_admit_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); } } admit_timer{e};
    if (n <= 0) return;
    Model& m = e.m;
    e.lock_model();
    Batch& b = *e.b;
    const Desc& d = m.d;
    hipStream_t s = e.io;   // beside the step chain, which keeps running on m.stream for the slots that are generating
    const int turn = e.arena_turn;
    e.arena_turn ^= 1;
    UploadScope upload_scope(e.arenas[turn], e.arena_free[turn]);   // the copies queued from this arena two admissions ago are long done
    const int B = e.B, D = d.d_model, ld = d.ldim;
    std::vector<int> slot_of((size_t)n);
    {
        int next = 0;
        for (int i = 0; i < n; i++) {
            while (next < B && e.slots[(size_t)next].busy) next++;
            if (next >= B) throw Error(PTTS_EINVAL, "continuous batch: more requests admitted than free slots");
            slot_of[(size_t)i] = next++;
        }
    }
    // voices: a free slot forgets its previous prefix first
    std::map<const Voice*, std::vector<int32_t>> by_voice;
    for (int i = 0; i < n; i++) {
        const int sl = slot_of[(size_t)i];
        b.kv_len_host[(size_t)sl] = 0;
        b.pre_k_host[(size_t)sl] = nullptr; b.pre_v_host[(size_t)sl] = nullptr; b.pre_len_host[(size_t)sl] = 0;
    }
    for (int i = 0; i < n; i++) {
        const ptts_request& r = *reqs[i];
        const int sl = slot_of[(size_t)i];
        if (r.voice) by_voice[reinterpret_cast<const Voice*>(r.voice)].push_back(sl);
        else if (r.voice_caches) batch_set_voice(b, sl, r.voice_caches, r.voice_cache_steps, r.voice_offsets);
    }
    for (auto& kv : by_voice) batch_apply_voice(b, *kv.first, kv.second);
    // prompt rows of the newcomers, packed; the running slots have empty segments
    std::vector<int64_t> row_off((size_t)B + 1, 0);
    std::vector<int> req_of_slot((size_t)B, -1);
    for (int i = 0; i < n; i++) req_of_slot[(size_t)slot_of[(size_t)i]] = i;
    for (int sl = 0; sl < B; sl++) {
        const int i = req_of_slot[(size_t)sl];
        const int64_t tp = i < 0 ? 0 : reqs[i]->n_tokens + (reqs[i]->voice_embedding ? reqs[i]->voice_frames : 0);
        row_off[(size_t)sl + 1] = row_off[(size_t)sl] + tp;
    }
    const int64_t R = row_off[(size_t)B];
    DevBuf& rows = m.work(5, (size_t)R * D * sizeof(float));
    {
        std::vector<int64_t> ids;
        for (int sl = 0; sl < B; sl++) {
            const int i = req_of_slot[(size_t)sl];
            if (i >= 0) ids.insert(ids.end(), reqs[i]->tokens, reqs[i]->tokens + reqs[i]->n_tokens);
        }
        DevBuf& dids = m.work(6, ids.size() * sizeof(int64_t));
        h2d(dids.p, ids.data(), ids.size() * sizeof(int64_t), s);
        int64_t id0 = 0;
        for (int sl = 0; sl < B; sl++) {   // (text embeddings and voice embeddings as GenerateAudio concatenates them, :89-119)
            const int i = req_of_slot[(size_t)sl];
            if (i < 0) continue;
            const ptts_request& r = *reqs[i];
            float* dst = rows.as<float>() + row_off[(size_t)sl] * D;
            const int64_t tv = r.voice_embedding ? r.voice_frames : 0;
            if (tv) h2d(dst, r.voice_embedding, (size_t)tv * D * sizeof(float), s);
            launch_embed_gather(m.at<float>(d.embed), dids.as<int64_t>() + id0, (int)r.n_tokens, D, dst + tv * D, s);
            id0 += r.n_tokens;
        }
    }
    batch_prompt(b, rows.as<float>(), row_off.data());
    // bookkeeping and noise rows of the new slots
    std::vector<SlotAdmit> adm((size_t)n);
    std::vector<NoiseSpec> spec((size_t)B, NoiseSpec{0, 0.0f, 0});
    bool any_draw = false;
    int draw_rows = 0;
    for (int i = 0; i < n; i++) {
        const ptts_request& r = *reqs[i];
        const int sl = slot_of[(size_t)i];
        const int ms = resolve_max_steps(r);
        adm[(size_t)i] = SlotAdmit{sl, ms, r.frames_after_eos, r.eos_threshold, b.kv_len_host[(size_t)sl], b.pre_len_host[(size_t)sl], b.pre_k_host[(size_t)sl],
                                   b.pre_v_host[(size_t)sl]};
        float* nrow = b.noise.as<float>() + (size_t)sl * b.max_steps * ld;
        PTTS_HIP(hipMemsetAsync(nrow, 0, (size_t)b.max_steps * ld * sizeof(float), s));
        if (r.noise) h2d(nrow, r.noise, (size_t)ms * ld * sizeof(float), s);
        else if (r.temperature > 0.0f) {
            if (ld % 4) throw Error(PTTS_EINVAL, "ptts-hip: the device noise draw needs a latent width that is a multiple of 4");
            spec[(size_t)sl] = NoiseSpec{r.noise_seed ? r.noise_seed : m.next_noise_seed(), std::sqrt(r.temperature), ms};
            any_draw = true;
            draw_rows = std::max(draw_rows, ms);
        }
    }
    if (any_draw) {
        DevBuf& sb = m.work(12, spec.size() * sizeof(NoiseSpec));
        h2d(sb.p, spec.data(), spec.size() * sizeof(NoiseSpec), s);
        launch_noise_fill(sb.as<NoiseSpec>(), B, draw_rows, b.noise.as<float>(), (int64_t)b.max_steps * ld, ld, s);
    }
    ContEngine::Joining j;
    j.ring = e.adm_turn;
    e.adm_turn = (e.adm_turn + 1) & 3;
    h2d(e.adm_dev[j.ring].p, adm.data(), adm.size() * sizeof(SlotAdmit), s);
    for (int i = 0; i < n; i++) {
        ContEngine::Slot& sl = e.slots[(size_t)slot_of[(size_t)i]];
        sl = ContEngine::Slot{};
        sl.busy = true; sl.req = reqs[i]; sl.res = results[i]; sl.tag = tags[i];
        sl.ms = adm[(size_t)i].max_steps;
        sl.base_kv = b.kv_len_host[(size_t)slot_of[(size_t)i]];
        sl.joining = true;
        j.slots.push_back(slot_of[(size_t)i]);
    }
    j.ready = e.event();
    j.seq = e.seq;
    PTTS_HIP(hipEventRecord(j.ready, s));
    PTTS_HIP(hipEventRecord(e.arena_free[turn], s));   // (the requests' own host arrays stay valid until their callers are woken)
    e.joining.push_back(std::move(j));
    e.admissions++;
    e.admitted += n;
    e.tr_admitted += n;
    e.last_admit_seq = e.seq;
}

// newcomers whose prefill has completed start stepping with the next group (activation is one small kernel on the step stream)
static void activate_ready(ContEngine& e, bool wait) {
    Batch& b = *e.b;
    while (!e.joining.empty()) {
        ContEngine::Joining& j = e.joining.front();
        if (wait) PTTS_HIP(hipEventSynchronize(j.ready));
        else {
            hipError_t q = hipEventQuery(j.ready);
            if (q == hipErrorNotReady) {
                // the prefill is still running beside the group in flight: the step STREAM waits for it (in front of the next group) rather than the host
                // looking again a group later -- a newcomer idles one group less, at the price of a short stall of everyone when the prefill is the slower
                // (waiting a group instead: occupancy 54.5 -> 58 of 64 slots with it, +3..5 % throughput).  The prefill stream is the model's second one, created
                // with the LOW priority (its one-shot job is the decoder behind the AR loop); a stream of its own at the step chain's priority would be the
                // engine's fourth, which measured 6.1 k x instead of 10.9 k (cont_create) -- the stall is bounded by one prefill (0.3-1.2 ms), not by the decoder's
                // queue.  The wait binds to the record made BEFORE this call (hipStreamWaitEvent captures the event's current record): handing j.ready back to
                // free_events below, and recording it again for a later prefill, does not move this wait.
                PTTS_HIP(hipStreamWaitEvent(e.m.stream, j.ready, 0));
            } else if (q != hipSuccess) throw Error(PTTS_ENODEVICE, strfmt("hip: hipEventQuery failed: %s", hipGetErrorString(q)));
        }
        launch_slot_admit(b.st, b.pre_len.as<int32_t>(), b.pre_k.as<const void*>(), b.pre_v.as<const void*>(), e.adm_dev[j.ring].as<SlotAdmit>(), (int)j.slots.size(),
                          e.m.stream);
        b.opened = false;   // a newcomer's first input is BOS: the next step is opened by k_step_begin, not by the previous step's last launch
        for (int sl : j.slots) { e.slots[(size_t)sl].joining = false; e.slots[(size_t)sl].admit_seq = e.seq; e.slots[(size_t)sl].seen_seq = e.seq; }
        e.n_gen += (int)j.slots.size();
        e.free_events.push_back(j.ready);
        e.joining.pop_front();
    }
}

// ended utterances -> audio, on the second stream, from their staging rows (which return to the pool when the frames have been packed)
static void start_decode(ContEngine& e, std::vector<ContEngine::Staged>& fin) {
    Model& m = e.m;
    Batch& b = *e.b;
    const Desc& d = m.d;
    const int ld = d.ldim;
    const int64_t spf = d.samples_per_frame;
    // Where the decode runs.  A handful of finished utterances (mixed lengths: ~16 at a time) goes to the confined stream BESIDE the following steps: such a decode
    // cannot fill the chip anyway and the steps it slows by 1.4-2 x are few.  A WAVE of them (uniform traffic: a whole engine's worth ends within a group or two) is
    // several full-chip decodes' worth of work; beside it every step would run on half of the CUs for its whole length -- it is queued in the step chain instead, on
    // all CUs, and the steps resume at full speed behind it (256 slots, uniform 10-s requests: 17.8 k -> 20.0 k x; mixed 2-12 s through the same rule: unchanged,
    // always-serial 14.6 k -> 12.5 k x; profiles/r5_serve_sweep.txt).
    static const int serial_frames = [] { const char* v = getenv("PTTS_CONT_SERIAL_FRAMES"); return v ? atoi(v) : kDecodeSerialFrames; }();
    int64_t fin_frames = 0;
    for (const ContEngine::Staged& f : fin) fin_frames += f.nf;
    const bool serial = fin_frames >= serial_frames;
    hipStream_t s = m.stream, s2 = serial ? m.stream : e.dec;
    const auto th0 = std::chrono::steady_clock::now();
    e.tr_decodes += (int)fin.size();
    std::sort(fin.begin(), fin.end(), [](const ContEngine::Staged& x, const ContEngine::Staged& y) { return x.nf > y.nf; });   // like lengths together: less padding
    // sub-groups whose decode fits the workspace; their frames are gathered FIRST, all of them, and on the AR stream itself: the staging rows are that
    // stream's (written by its copies, refilled by its copies: no other stream ever has to hand them back), and the gather is a few microseconds there.
    // (Round 3 gathered on the decoder's stream and made the AR stream wait for the LAST sub-group's gather -- which sat behind the first sub-group's whole
    // decode: every decode of more than kFramesPerDecode frames stopped the step chain for ~5 ms, a quarter of the engine's time; PTTS_CONT_TRACE.)
    struct Sub { size_t at, end; int T; size_t off; };
    std::vector<Sub> subs;
    size_t total = 0;
    for (size_t at = 0; at < fin.size();) {
        size_t end = at;
        int T = 0;
        while (end < fin.size()) {
            const int t2 = std::max(T, std::max(1, fin[end].nf));
            if (end > at && (int64_t)t2 * (int64_t)(end - at + 1) > kFramesPerDecode) break;
            // (sorted by length: everything in the sub-group is padded to its first; a member much shorter than that starts the next one)
            // (without this rule a third of the decoder's frames were padding: 74 964 decoded for 56 948 real ones in the benchmark run; with it 60 757)
            if (end > at && fin[at].nf - fin[end].nf > std::max(kBandFrames, fin[at].nf * kBandPercent / 100)) break;
            T = t2; end++;
        }
        subs.push_back(Sub{at, end, T, total});
        total += (end - at) * (size_t)T * ld;
        e.tr_padded += (int64_t)(end - at) * T; e.tr_subs++;
        for (size_t i = at; i < end; i++) e.tr_frames += fin[i].nf;
        at = end;
    }
    // (four buffers in rotation: with two, the step stream waited here for the decode BEFORE LAST to end whenever the decoder's stream had a backlog -- which
    // on mixed traffic it has most of the time: 2.5 ms in front of every group that followed a decode start, PTTS_CONT_TRACE)
    const int lt = e.lat_turn, lprev = (lt + ContEngine::kLat - 1) % ContEngine::kLat;
    e.lat_turn = (lt + 1) % ContEngine::kLat;
    DevBuf& lat_all = e.lat[lt];
    if (e.lat_busy[lt]) {   // the decode four starts ago has read this buffer -- and the host rewrites that decode's row table below: it has been uploaded
        PTTS_HIP(hipStreamWaitEvent(s, e.lat_free[lt], 0));
        PTTS_HIP(hipEventSynchronize(e.lat_free[lt]));
    }
    lat_all.ensure(total * sizeof(float));
    {   // one gather launch per 128 utterances (its table rides in the kernel arguments) instead of a memset and a copy per utterance
        GatherTable tab;
        int nt = 0;
        for (const Sub& sb : subs)
            for (size_t i = sb.at; i < sb.end; i++) {
                tab.rows[nt++] = GatherTable::Row{(int32_t)fin[i].row, (int32_t)(sb.off + (i - sb.at) * (size_t)sb.T * ld), (int32_t)std::max(0, fin[i].nf), (int32_t)sb.T};
                if (nt == 128) { launch_gather_frames(tab, nt, e.stage.as<float>(), (int64_t)b.max_steps * ld, lat_all.as<float>(), ld, s); nt = 0; }
            }
        launch_gather_frames(tab, nt, e.stage.as<float>(), (int64_t)b.max_steps * ld, lat_all.as<float>(), ld, s);
    }
    const auto th1 = std::chrono::steady_clock::now();
    if (s2 != s) {
        PTTS_HIP(hipEventRecord(e.ev_steps, s));
        PTTS_HIP(hipStreamWaitEvent(s2, e.ev_steps, 0));
    }
    if (e.lat_busy[lprev]) PTTS_HIP(hipStreamWaitEvent(s2, e.lat_free[lprev], 0));   // the previous decode -- possibly on the other stream -- has left the decoder's workspace
    size_t n_rows_used = 0;
    for (const Sub& sb : subs) {
        const size_t at = sb.at, end = sb.end;
        const int T = sb.T;
        const int nb = (int)(end - at);
        float* const lat = lat_all.as<float>() + sb.off;
        MimiWs mw;
        mimi_setup(m, mw, nb, T);
        mimi_zero_history(m, mw, s2);
        DevBuf& pcm = m.work(7, (size_t)nb * T * spf * sizeof(float));
        // The result rows first: the decoder's last kernel stores every utterance's samples (f32 or int16) straight into its page-locked result buffer -- the kernel's
        // stores ARE the device -> host transfer, as in the one-shot path (runtime.cpp generate_chunk): no PCM copy per utterance on the decoder's stream, no
        // conversion launch.  (A pool that had to fall back to pageable memory, or a decoder shape the fused last stage does not take: the buffer + copy path below.)
        const size_t row0 = (size_t)lt * 2 * (size_t)e.B + n_rows_used;
        PcmRow* hrows = e.rows_host + row0;
        bool direct = row0 + (size_t)nb <= (size_t)ContEngine::kLat * 2 * (size_t)e.B && n_rows_used + (size_t)nb <= 2 * (size_t)e.B;
        for (int i = 0; i < nb; i++) {
            const ContEngine::Staged& f = fin[at + (size_t)i];
            ptts_result& r = *f.res;
            r.n_frames = f.nf; r.eos_step = f.eos; r.n_samples = (int64_t)f.nf * spf; r.status = PTTS_OK;
            const bool s16r = f.req->pcm_format == PTTS_PCM_S16;
            void* dst = result_alloc((size_t)std::max<int64_t>(1, r.n_samples) * (s16r ? sizeof(int16_t) : sizeof(float)));
            if (s16r) r.pcm16 = (int16_t*)dst; else r.pcm = (float*)dst;
            if (!dst) { r.status = PTTS_ENOMEM; direct = false; continue; }
            direct = direct && result_is_pinned(dst);
            if (direct) hrows[i] = PcmRow{dst, (int32_t)std::min<int64_t>(r.n_samples, INT32_MAX), s16r ? 1 : 0};
        }
        const PcmRow* drows = nullptr;
        if (direct) {
            PcmRow* dr = e.rows_dev.as<PcmRow>() + row0;
            PTTS_HIP(hipMemcpyAsync(dr, hrows, (size_t)nb * sizeof(PcmRow), hipMemcpyHostToDevice, s2));   // (page-locked source, rewritten no sooner than kLat decode starts later)
            drows = dr;
            n_rows_used += (size_t)nb;
        }
        bool stored = false;
        mimi_range(m, mw, lat, (int64_t)T * ld, 0, T, pcm.as<float>(), nullptr, s2, drows, &stored);
        bool any_s16 = false;
        for (int i = 0; i < nb; i++) any_s16 |= fin[at + (size_t)i].req->pcm_format == PTTS_PCM_S16;
        DevBuf* s16 = nullptr;
        if (!stored && any_s16) {
            s16 = &m.work(8, (size_t)nb * T * spf * sizeof(int16_t));
            launch_pcm16(pcm.as<float>(), s16->as<int16_t>(), (int64_t)nb * T * spf, s2);
        }
        ContEngine::Pending p;
        for (int i = 0; i < nb; i++) {
            const ContEngine::Staged& f = fin[at + (size_t)i];
            ptts_result& r = *f.res;
            const int nf = f.nf;
            if (!stored && r.status == PTTS_OK && r.n_samples > 0) {
                if (f.req->pcm_format == PTTS_PCM_S16)
                    PTTS_HIP(hipMemcpyAsync(r.pcm16, s16->as<int16_t>() + (size_t)i * T * spf, (size_t)r.n_samples * sizeof(int16_t), hipMemcpyDeviceToHost, s2));
                else
                    PTTS_HIP(hipMemcpyAsync(r.pcm, pcm.as<float>() + (size_t)i * T * spf, (size_t)r.n_samples * sizeof(float), hipMemcpyDeviceToHost, s2));
            }
            if (f.req->want_latents && r.status == PTTS_OK) {
                r.latents = (float*)malloc((size_t)std::max(1, nf) * ld * sizeof(float));
                if (!r.latents) r.status = PTTS_ENOMEM;
                else if (nf > 0)
                    PTTS_HIP(hipMemcpyAsync(r.latents, lat + (size_t)i * T * ld, (size_t)nf * ld * sizeof(float), hipMemcpyDeviceToHost, s2));
            }
            p.tags.push_back(f.tag);
            p.res.push_back(f.res);
            e.stage_free.push_back(f.row);
        }
        p.done = e.event();
        PTTS_HIP(hipEventRecord(p.done, s2));
        e.decoding.push_back(std::move(p));
    }
    PTTS_HIP(hipEventRecord(e.lat_free[lt], s2));
    e.lat_busy[lt] = true;
    e.tr_starts++;
    e.tr_host_gather_us += std::chrono::duration<double, std::micro>(th1 - th0).count();
    e.tr_host_launch_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - th1).count();
    e.tr_acc_decode_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - th0).count();
    fin.clear();
}

// what one read-back says about the slots it covers: ended utterances are marked (their frames wait for the decoder), cancelled ones are
// answered at once and switched off
static void take_snapshot(ContEngine& e, ContEngine::Snap& sn, std::vector<void*>& done) {
    Model& m = e.m;
    Batch& b = *e.b;
    const int B = e.B;
    {
        const auto tw = std::chrono::steady_clock::now();
        PTTS_HIP(hipEventSynchronize(sn.ready));
        e.tr_acc_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw).count();
    }
    sn.pending = false;
    if (sn.host[5 * (size_t)B]) { sn.host[5 * (size_t)B] = 0; flow_cluster_fault(b); }   // throws: the dispatcher fails everyone in flight and rebuilds the engine
    const int32_t* active = sn.host;
    const int32_t* n_frames = sn.host + 3 * (size_t)B;
    const int32_t* eos_step = sn.host + 4 * (size_t)B;
    std::vector<int32_t> retire;
    // the frames of the utterances that ended move to their staging rows in ONE launch (a copy per utterance held the step stream for ~100 us each)
    const int ld = m.d.ldim;
    GatherTable moves;
    int n_moves = 0;
    auto flush_moves = [&] {
        launch_gather_frames(moves, n_moves, b.latents.as<float>(), (int64_t)b.max_steps * ld, e.stage.as<float>(), ld, m.stream);
        n_moves = 0;
    };
    for (int sl = 0; sl < B; sl++) {
        ContEngine::Slot& so = e.slots[(size_t)sl];
        if (!so.busy || so.joining || sn.seq <= so.admit_seq) continue;   // (a read-back from before the slot was filled says nothing about it)
        so.nf_seen = n_frames[sl]; so.seen_seq = sn.seq;
        if (so.req->cancel && *so.req->cancel) {   // ctx.Err() between steps (runtime_native_safetensors.go:156-159)
            retire.push_back(sl);
            so.res->status = PTTS_ECANCELLED;
            done.push_back(so.tag);
            so = ContEngine::Slot{};
            e.n_gen--;
        } else if (!active[sl]) {   // ended: its frames move to a staging row (queued behind the group in flight), the slot is free
            const int nf = n_frames[sl];
            if (e.stage_free.empty()) { flush_moves(); start_decode(e, e.staged); }   // (cannot run dry before this: 2 B rows, at most B slots + what one turn adds)
            const int row = e.stage_free.back();
            e.stage_free.pop_back();
            if (nf > 0) {
                moves.rows[n_moves++] = GatherTable::Row{(int32_t)sl, (int32_t)((int64_t)row * b.max_steps * ld), (int32_t)nf, (int32_t)nf};
                if (n_moves == 128) flush_moves();
            }
            e.staged.push_back(ContEngine::Staged{so.req, so.res, so.tag, nf, eos_step[sl], row, e.seq});
            so = ContEngine::Slot{};
            e.n_gen--;
        }
    }
    flush_moves();
    if (!retire.empty()) {   // queued behind the group in flight; whatever refills the slot is queued behind this
        const int turn = e.arena_turn;
        e.arena_turn ^= 1;
        UploadScope upload_scope(e.arenas[turn], e.arena_free[turn]);
        DevBuf& dr = m.work(15, (size_t)B * sizeof(int32_t));
        h2d(dr.p, retire.data(), retire.size() * sizeof(int32_t), m.stream);
        launch_slot_retire(b.st, dr.as<int32_t>(), (int)retire.size(), m.stream);
        PTTS_HIP(hipEventRecord(e.arena_free[turn], m.stream));
    }
}

// One turn of the engine: a group of `steps` AR steps is queued (if anything is generating) and its read-back behind it; the read-back of
// the PREVIOUS group -- complete by now -- is looked at; finished utterances are handed to the decoder a handful at a time; `done`
// receives the tags whose results are complete (status set).  drain: do not return before everything in flight has been answered.
void cont_advance(ContEngine& e, int steps, std::vector<void*>& done, bool drain) {
    Model& m = e.m;
    if (e.busy() == 0) return;
    e.lock_model();
    Batch& b = *e.b;
    const int B = e.B;
    hipStream_t s = m.stream;
    steps = std::max(1, steps);
    e.group_steps = steps;
    do {
        activate_ready(e, e.n_gen == 0);   // (nothing generating: the newcomers are all there is to wait for)
        if (e.n_gen > 0) {
            // upper bound on any live slot's cache length when this group runs (the step attention issues load rounds by it)
            int bound = 0;
            for (int sl = 0; sl < B; sl++) {
                const ContEngine::Slot& so = e.slots[(size_t)sl];
                if (so.busy && !so.joining) bound = std::max(bound, so.base_kv + std::min(so.ms, so.nf_seen + steps * (int)(e.seq - so.seen_seq)));
            }
            b.kv_bound = std::min(bound, e.cap);
            ContEngine::GroupRec rec{};
            if (e.trace_path) {
                PTTS_HIP(hipEventCreate(&rec.t0)); PTTS_HIP(hipEventCreate(&rec.t1));
                rec.n_gen = e.n_gen; rec.steps = steps; rec.admitted = e.tr_admitted; rec.decodes_started = e.tr_decodes; rec.decoding = (int)e.decoding.size(); rec.joining = (int)e.joining.size();
                const auto tn = std::chrono::steady_clock::now();
                rec.host_turn_us = (float)std::chrono::duration<double, std::micro>(tn - e.tr_last_enqueue).count();
                rec.host_admit_us = (float)e.tr_acc_admit_us; rec.host_decode_us = (float)e.tr_acc_decode_us; rec.host_wait_us = (float)e.tr_acc_wait_us;
                e.tr_last_enqueue = tn; e.tr_acc_admit_us = e.tr_acc_decode_us = e.tr_acc_wait_us = 0;
                e.tr_admitted = e.tr_decodes = 0;
                PTTS_HIP(hipEventRecord(rec.t0, s));
            }
            if (e.use_graph) enqueue_step(b, 1, true, steps);
            else for (int k = 0; k < steps; k++) enqueue_step(b, 1, false, 1);
            if (e.trace_path) { PTTS_HIP(hipEventRecord(rec.t1, s)); e.trace.push_back(rec); }
            e.seq++;
            e.steps_run += steps; e.slot_steps += (int64_t)steps * e.n_gen;
            ContEngine::Snap& sn = e.snaps[e.seq & 1];
            if (sn.pending) take_snapshot(e, sn, done);   // (cannot happen: the older read-back is consumed every turn)
            launch_readback_i32(b.st.active, 5 * B, b.fc_ok ? reinterpret_cast<const int32_t*>(b.fc_fault()) : nullptr, 1, sn.host, s);   // (a kernel, not the copy engine: kernels.hip)
            PTTS_HIP(hipEventRecord(sn.ready, s));
            sn.seq = e.seq; sn.pending = true;
        }
        // decode: a good handful at a time (small decodes are inefficient and disturb the step chain as much as large ones), or whatever
        // there is once the oldest has waited eight groups / nothing generates / the caller drains
        auto decode_if_due = [&] {
            if (e.staged.empty()) return;
            uint64_t oldest = e.seq;
            for (const ContEngine::Staged& f : e.staged) oldest = std::min(oldest, f.seq);
            if ((int)e.staged.size() >= std::min(kDecodeMin, std::max(4, B / 3)) || (int)(e.seq - oldest) >= kDecodeMaxAge || e.n_gen == 0 || drain) start_decode(e, e.staged);
        };
        // While steps are running the host is about to WAIT for the older read-back (the group before the one just queued is still on the GPU: ~1.3 ms at 192 slots):
        // the decoder's ~50-200 launches (0.8 ms of host time per start) go in front of that wait, for what earlier turns staged -- behind it they sat on the critical
        // path of the next group's enqueue, and the step stream stood still for ~1.3 ms in front of every group that followed a decode start (PTTS_CONT_TRACE)
        ContEngine::Snap& older = e.snaps[(e.seq + 1) & 1];
        const bool overlap = older.pending && e.n_gen > 0 && !drain;
        if (overlap) decode_if_due();
        // the older read-back first; the newest as well when nothing else will be queued behind it
        if (older.pending) take_snapshot(e, older, done);
        ContEngine::Snap& newest = e.snaps[e.seq & 1];
        if (newest.pending && (drain || e.n_gen == 0)) take_snapshot(e, newest, done);
        if (!overlap || e.n_gen == 0) decode_if_due();
        while (!e.decoding.empty()) {
            ContEngine::Pending& p = e.decoding.front();
            if (drain || (e.n_gen == 0 && e.n_finished() == 0)) PTTS_HIP(hipEventSynchronize(p.done));
            else {
                hipError_t q = hipEventQuery(p.done);
                if (q == hipErrorNotReady) break;
                if (q != hipSuccess) throw Error(PTTS_ENODEVICE, strfmt("hip: hipEventQuery failed: %s", hipGetErrorString(q)));
            }
            for (void* t : p.tags) done.push_back(t);
            e.free_events.push_back(p.done);
            p.done = nullptr;
            e.decoding.pop_front();
        }
    } while (drain && e.busy() > 0);
    e.unlock_if_idle();
}

// newcomers are admitted whenever a slot is free: every turn of the engine (round 3 admitted every fourth group to spare the step chain the prefills; since
// small prefills run as chunks of the step kernel -- kernels.hip launch_gemm -- a prefill costs the chain ~7 % while it runs, an idle slot costs 1.6 % each)
bool cont_admit_now(const ContEngine& e, int waiting) {
    return waiting > 0 && e.free_slots() > 0;
}

// every request in flight is answered with `code` (the engine is about to be torn down after an error)
void cont_abort(ContEngine& e, int code, std::vector<void*>& done) {
    (void)hipStreamSynchronize(e.m.stream);
    (void)hipStreamSynchronize(e.m.stream2);
    if (e.dec) (void)hipStreamSynchronize(e.dec);
    if (e.io) (void)hipStreamSynchronize(e.io);
    for (auto& j : e.joining) if (j.ready) e.free_events.push_back(j.ready);
    e.joining.clear();
    for (ContEngine::Snap& sn : e.snaps) sn.pending = false;
    for (auto& so : e.slots) {
        if (!so.busy) continue;
        so.res->status = code;
        done.push_back(so.tag);
        so = ContEngine::Slot{};
    }
    e.n_gen = 0;
    for (auto& f : e.staged) { f.res->status = code; done.push_back(f.tag); e.stage_free.push_back(f.row); }
    e.staged.clear();
    for (auto& p : e.decoding) {
        for (size_t i = 0; i < p.tags.size(); i++) { ptts_free_result(p.res[i]); p.res[i]->status = code; p.res[i]->eos_step = -1; done.push_back(p.tags[i]); }
        if (p.done) { e.free_events.push_back(p.done); p.done = nullptr; }
    }
    e.decoding.clear();
    e.unlock_if_idle();
}

}  // namespace ptts
