// dispatcher.cpp -- SURVEY.md 8f N1: what the reference's worker pool becomes in front of a batching GPU runtime.
//
// The reference admits `workers` (default 2) concurrent Synthesize calls through a channel semaphore and lets the rest wait
// (internal/server/server.go:132-134,398-421); each admitted call then runs its chunks one by one, and the native runtime
// serialises the AR steps of concurrent calls on a mutex (runtime_native_safetensors.go:161-170).  On the GPU a step costs
// the same for 1 or 64 utterances, so the unit of admission is a BATCH: callers block in ptts_dispatch_generate, a worker
// thread per model (one model per GPU) takes the oldest waiting request, keeps collecting for at most `window_us` (or until
// `max_batch`), runs ONE ptts_generate-equivalent for the lot and wakes each caller with its own result.  A request whose
// cancel flag is raised while it waits is answered PTTS_ECANCELLED without running, the counterpart of "request cancelled
// while waiting for worker" (server.go:415-418).  Utterance chunks are independent, so nothing is exchanged between GPUs:
// several models simply pull from the same queue.
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>

#include "runtime.h"

namespace ptts {

const std::string& last_error_ref();

using Clock = std::chrono::steady_clock;

struct DispatchItem {
    const ptts_request* req = nullptr;
    ptts_result* res = nullptr;
    const Voice* voice = nullptr;       // a device-resident voice ties the request to the models that can read it (same GPU)
    Clock::time_point enq;
    int rc = PTTS_OK;
    std::string err;
    bool done = false;
    std::condition_variable cv;
};

struct Dispatcher {
    std::vector<Model*> models;         // empty for a custom executor (CPU tests of the queueing logic)
    ExecFn exec = nullptr;
    void* exec_user = nullptr;
    int n_workers = 0;
    int max_batch = 64;
    int window_us = 2000;
    int queue_cap = 4096;
    std::mutex mu;
    std::condition_variable cv_work;
    std::deque<DispatchItem*> queue;
    bool closing = false;
    Clock::time_point last_arrival = Clock::now();
    Clock::time_point last_batch_left = Clock::now();
    bool collecting = false;            // one worker at a time forms a batch; the others wait their turn
    std::vector<std::thread> workers;
    // continuous batching (continuous.cpp): one long-lived batch per model; free slots are refilled between groups of AR steps
    bool continuous = false;
    int cont_kv_cap = 512, cont_max_steps = 256, cont_group = 3;
    // statistics
    int64_t n_requests = 0, n_batches = 0, n_cancelled_waiting = 0, max_depth = 0;
    double sum_wait_us = 0.0, sum_exec_us = 0.0;
    int64_t cont_steps = 0, cont_slot_steps = 0;   // continuous engines: AR steps launched, utterances stepping in them (summed)

    void run(int w);
    void run_continuous(int w);
    void run_batch(Model* model, int w, std::vector<DispatchItem*>& batch);
};

static bool cancelled(const DispatchItem* it) { return it->req->cancel && *it->req->cancel; }

void Dispatcher::run(int w) {
    Model* model = models.empty() ? nullptr : models[(size_t)w];
    if (model) model->use_device();
    for (;;) {
        std::vector<DispatchItem*> batch;
        {
            std::unique_lock<std::mutex> lock(mu);
            auto eligible = [&](const DispatchItem* it) { return !it->voice || !model || voice_usable_by(*it->voice, *model); };
            auto first = [&]() -> DispatchItem* {
                for (DispatchItem* it : queue) if (eligible(it)) return it;
                return nullptr;
            };
            cv_work.wait(lock, [&] { return closing || (!collecting && first() != nullptr); });
            if (closing && first() == nullptr) return;
            collecting = true;
            // coalescing window: counted from the moment the oldest eligible request arrived, so a lone request waits at
            // most window_us and a full batch leaves at once
            // ... and it stretches (to at most four windows) while requests are still arriving: a burst of callers -- the
            // chunks of one long text, clients released by the previous batch -- trickles in over a few milliseconds, and
            // cutting it in two costs a whole extra batch time (tools/serve_bench.py: 64 clients 5.2 k -> 7.7 k x real time)
            // what a departing batch left behind gets a fresh window: the callers it is about to release will join it
            const Clock::time_point t_first = std::max(first()->enq, last_batch_left);
            const auto window = std::chrono::microseconds(window_us), quiet = std::chrono::microseconds(std::max(1, window_us / 4));
            auto n_eligible = [&] { int n = 0; for (DispatchItem* it : queue) n += eligible(it); return n; };
            for (;;) {
                if (closing || n_eligible() >= max_batch) break;
                const Clock::time_point now = Clock::now();
                Clock::time_point deadline = t_first + window;
                if (now >= deadline) {
                    if (now - last_arrival >= quiet || now >= t_first + 4 * window) break;
                    deadline = std::min(last_arrival + quiet, t_first + 4 * window);
                }
                cv_work.wait_until(lock, deadline);
            }
            const Clock::time_point now = Clock::now();
            for (auto it = queue.begin(); it != queue.end() && (int)batch.size() < max_batch;) {
                DispatchItem* d = *it;
                if (!eligible(d)) { ++it; continue; }
                it = queue.erase(it);
                if (cancelled(d)) {   // never ran: the reference answers 503 "request cancelled while waiting for worker"
                    d->rc = PTTS_ECANCELLED;
                    d->err = "request cancelled while waiting for worker";
                    d->done = true;
                    n_cancelled_waiting++;
                    d->cv.notify_one();
                    continue;
                }
                sum_wait_us += std::chrono::duration<double, std::micro>(now - d->enq).count();
                batch.push_back(d);
            }
            collecting = false;
            last_batch_left = now;
            cv_work.notify_all();
            if (batch.empty()) continue;
            n_batches++;
            n_requests += (int64_t)batch.size();
        }
        run_batch(model, w, batch);
    }
}

// one batched generate for the collected requests; every caller is woken with its own result
void Dispatcher::run_batch(Model* model, int w, std::vector<DispatchItem*>& batch) {
    {
        std::vector<ptts_request> reqs(batch.size());
        std::vector<ptts_result> ress(batch.size());
        for (size_t i = 0; i < batch.size(); i++) reqs[i] = *batch[i]->req;
        std::string err;
        int rc = PTTS_OK;
        const Clock::time_point t0 = Clock::now();
        if (exec) {
            char buf[256] = {0};
            for (auto& r : ress) { std::memset(&r, 0, sizeof r); r.eos_step = -1; }
            rc = exec(exec_user, w, reqs.data(), (int32_t)reqs.size(), ress.data(), buf, (int32_t)sizeof buf);
            err = buf;
        } else {
            try {
                set_last_error("");
                generate(*model, reqs.data(), (int)reqs.size(), ress.data());
                err = last_error_ref();
            } catch (const Error& e) {
                rc = e.code; err = e.what();
            } catch (const std::exception& e) {
                rc = PTTS_EINVAL; err = std::string("ptts-hip: ") + e.what();
            }
        }
        const double exec_us = std::chrono::duration<double, std::micro>(Clock::now() - t0).count();
        {
            std::lock_guard<std::mutex> lock(mu);
            sum_exec_us += exec_us;
            for (size_t i = 0; i < batch.size(); i++) {
                DispatchItem* d = batch[i];
                *d->res = ress[i];
                d->rc = rc != PTTS_OK ? rc : ress[i].status;
                if (d->rc != PTTS_OK) d->err = err.empty() ? "generate: request failed" : err;
                d->done = true;
                d->cv.notify_one();
            }
        }
    }
}

// Continuous batching: this worker owns one long-lived batch on its model.  Each turn: answer what was cancelled while waiting, move
// waiting requests the engine can take into its free slots (prefill + bookkeeping), run a group of AR steps, read the counters back,
// start the decoder on whatever ended, wake the callers whose audio has arrived.  Requests the engine cannot take (per-step callbacks,
// streaming, lsd_steps > 1, budgets beyond its geometry) are run batch-at-a-time whenever the engine is empty; once such a request has
// waited 50 ms the engine stops admitting until it has drained, so that it cannot starve.
void Dispatcher::run_continuous(int w) {
    Model* model = models[(size_t)w];
    model->use_device();
    std::unique_ptr<ContEngine, void (*)(ContEngine*)> eng(nullptr, cont_destroy);
    auto fail_all = [&](std::vector<void*>& tags, int code, const std::string& msg) {
        std::lock_guard<std::mutex> lock(mu);
        for (void* t : tags) {
            DispatchItem* d = static_cast<DispatchItem*>(t);
            d->rc = code; d->err = msg; d->res->status = code; d->done = true; d->cv.notify_one();
        }
    };
    try {   // the engine's buffers (KV caches, the decoder's workspace) exist before the first caller arrives
        eng.reset(cont_create(*model, std::min(max_batch, std::max(1, model->opts.max_batch)), cont_kv_cap, cont_max_steps));
    } catch (const std::exception&) {}   // (retried, and reported to the callers, in the loop)
    for (;;) {
        std::vector<DispatchItem*> take, classic;
        try {
            if (!eng) eng.reset(cont_create(*model, std::min(max_batch, std::max(1, model->opts.max_batch)), cont_kv_cap, cont_max_steps));
            const int busy = cont_busy(*eng);
            {
                std::unique_lock<std::mutex> lock(mu);
                auto eligible = [&](const DispatchItem* it) { return !it->voice || voice_usable_by(*it->voice, *model); };
                auto any_eligible = [&] { for (DispatchItem* it : queue) if (eligible(it)) return true; return false; };
                if (busy == 0) {
                    cv_work.wait(lock, [&] { return closing || any_eligible(); });
                    if (!any_eligible()) { if (closing) return; continue; }
                    // an idle engine gives a burst the same short window as the batch collector before it starts stepping
                    const Clock::time_point t_first = Clock::now();
                    const auto window = std::chrono::microseconds(window_us);
                    while (!closing && (int)queue.size() < max_batch && Clock::now() < t_first + window) cv_work.wait_until(lock, t_first + window);
                }
                const Clock::time_point now = Clock::now();
                bool classic_waiting_long = false;
                for (DispatchItem* it : queue)
                    if (eligible(it) && !cont_accepts(*eng, *it->req) && now - it->enq > std::chrono::milliseconds(50)) classic_waiting_long = true;
                // an empty engine serves the oldest waiting request in the way that request needs: batch-at-a-time if it cannot be taken
                bool classic_mode = false;
                if (busy == 0)
                    for (DispatchItem* it : queue)
                        if (eligible(it) && !cancelled(it)) { classic_mode = !cont_accepts(*eng, *it->req); break; }
                int n_fit = 0;
                for (DispatchItem* it : queue) n_fit += eligible(it) && !cancelled(it) && cont_accepts(*eng, *it->req);
                int room = (classic_waiting_long || classic_mode || !cont_admit_now(*eng, n_fit)) ? 0 : cont_free_slots(*eng);
                for (auto it = queue.begin(); it != queue.end();) {
                    DispatchItem* d = *it;
                    if (!eligible(d)) { ++it; continue; }
                    if (cancelled(d)) {
                        it = queue.erase(it);
                        d->rc = PTTS_ECANCELLED; d->err = "request cancelled while waiting for worker"; d->done = true;
                        n_cancelled_waiting++;
                        d->cv.notify_one();
                        continue;
                    }
                    const bool fits = cont_accepts(*eng, *d->req);
                    if (fits && room > 0) {
                        it = queue.erase(it); take.push_back(d); room--;
                        sum_wait_us += std::chrono::duration<double, std::micro>(now - d->enq).count();
                        continue;
                    }
                    if (!fits && classic_mode && (int)classic.size() < max_batch) {
                        it = queue.erase(it); classic.push_back(d);
                        sum_wait_us += std::chrono::duration<double, std::micro>(now - d->enq).count();
                        continue;
                    }
                    ++it;
                }
                if (!take.empty()) { n_batches++; n_requests += (int64_t)take.size(); }
                else if (!classic.empty()) { n_batches++; n_requests += (int64_t)classic.size(); }
            }
            if (take.empty() && !classic.empty()) { run_batch(model, w, classic); continue; }
            const Clock::time_point t0 = Clock::now();
            if (!take.empty()) {
                std::vector<const ptts_request*> rq; std::vector<ptts_result*> rs; std::vector<void*> tg;
                for (DispatchItem* d : take) { rq.push_back(d->req); rs.push_back(d->res); tg.push_back(d); }
                cont_admit(*eng, rq.data(), rs.data(), tg.data(), (int)take.size());
                take.clear();
            }
            std::vector<void*> done;
            int64_t st0 = 0, ss0 = 0, st1 = 0, ss1 = 0;
            cont_occupancy(*eng, &st0, &ss0);
            cont_advance(*eng, cont_group, done, false);
            cont_occupancy(*eng, &st1, &ss1);
            const double exec_us = std::chrono::duration<double, std::micro>(Clock::now() - t0).count();
            if (!done.empty() || exec_us > 0) {
                std::lock_guard<std::mutex> lock(mu);
                sum_exec_us += exec_us;
                cont_steps += st1 - st0; cont_slot_steps += ss1 - ss0;
                for (void* t : done) {
                    DispatchItem* d = static_cast<DispatchItem*>(t);
                    d->rc = d->res->status;
                    if (d->rc == PTTS_ECANCELLED) d->err = "context canceled";
                    else if (d->rc != PTTS_OK) d->err = "generate: request failed";
                    d->done = true;
                    d->cv.notify_one();
                }
            }
        } catch (const FlowClusterFault&) {
            // a hand-off inside k_flow_cluster timed out (its workgroups were not running together): nobody is failed.  Everything in flight goes back to the
            // head of the queue with a clean result, the engine is rebuilt -- on the 2 x depth launches from now on (Model::fc_disabled), which compute the
            // same bits -- and the utterances start over.  Counted in ptts_dispatch_stats.flow_cluster_fallbacks.
            std::vector<void*> tags;
            for (DispatchItem* d : take) tags.push_back(d);
            if (eng) cont_abort(*eng, PTTS_ENODEVICE, tags);
            eng.reset();
            std::lock_guard<std::mutex> lock(mu);
            for (auto it = tags.rbegin(); it != tags.rend(); ++it) {
                DispatchItem* d = static_cast<DispatchItem*>(*it);
                ptts_free_result(d->res);
                std::memset(d->res, 0, sizeof *d->res);
                d->res->eos_step = -1;
                queue.push_front(d);
            }
        } catch (const std::exception& ex) {   // a HIP error or a malformed request that slipped through: everyone in flight is told, the engine is rebuilt
            const Error* pe = dynamic_cast<const Error*>(&ex);
            const int code = pe ? pe->code : PTTS_EINVAL;
            std::vector<void*> tags;
            for (DispatchItem* d : take) tags.push_back(d);
            if (eng) cont_abort(*eng, code, tags);
            fail_all(tags, code, ex.what());
            eng.reset();
        }
    }
}

Dispatcher* dispatcher_create(Model* const* models, int n_models, ExecFn exec, void* user, int n_workers, int max_batch, int window_us, int queue_cap,
                              const DispatchCont* cont) {
    std::unique_ptr<Dispatcher> d(new Dispatcher());
    if (exec) {
        d->exec = exec; d->exec_user = user; d->n_workers = std::max(1, n_workers);
    } else {
        if (n_models <= 0 || !models) throw Error(PTTS_EINVAL, "dispatcher: at least one model is required");
        for (int i = 0; i < n_models; i++) {
            if (!models[i]) throw Error(PTTS_EINVAL, "dispatcher: nil model");
            d->models.push_back(models[i]);
        }
        d->n_workers = n_models;
        if (max_batch <= 0) max_batch = models[0]->opts.max_batch;
    }
    d->max_batch = std::max(1, max_batch <= 0 ? 64 : max_batch);
    d->window_us = std::max(0, window_us);
    d->queue_cap = queue_cap <= 0 ? 4096 : queue_cap;
    // continuous batching: 1 on, -1 off, 0 the library's choice -- on where it wins both kinds of traffic, i.e. when every model of the dispatcher has its GPU to
    // itself (one engine per GPU: uniform 10-s requests 19.8 k x against 18.5 k x for the best batch-at-a-time setting, mixed 2-12 s 14.4 k against 10.8 k,
    // profiles/r5_serve_sweep.txt); two engines sharing a GPU (ptts_model_share) are the batch-at-a-time setting: as continuous engines they would lose to one
    int on = cont ? cont->on : 0;
    if (on == 0 && !exec) {
        on = 1;
        for (size_t i = 0; i < d->models.size(); i++)
            for (size_t j = i + 1; j < d->models.size(); j++)
                if (d->models[i]->device == d->models[j]->device) on = -1;
    }
    if (cont && on > 0 && !exec) {
        d->continuous = true;
        if (cont->kv_capacity > 0) d->cont_kv_cap = cont->kv_capacity;
        if (cont->max_steps > 0) d->cont_max_steps = cont->max_steps;
        if (cont->steps_per_group > 0) d->cont_group = cont->steps_per_group;
    }
    for (int w = 0; w < d->n_workers; w++) d->workers.emplace_back([p = d.get(), w] { if (p->continuous) p->run_continuous(w); else p->run(w); });
    return d.release();
}

void dispatcher_close(Dispatcher* d) {
    if (!d) return;
    {
        std::lock_guard<std::mutex> lock(d->mu);
        d->closing = true;
    }
    d->cv_work.notify_all();
    for (auto& t : d->workers) t.join();
    {   // whatever is still queued (only possible for requests pinned to no live worker) is refused
        std::lock_guard<std::mutex> lock(d->mu);
        for (DispatchItem* it : d->queue) { it->rc = PTTS_ECANCELLED; it->err = "dispatcher closed"; it->done = true; it->cv.notify_one(); }
        d->queue.clear();
    }
    delete d;
}

int dispatcher_generate(Dispatcher* d, const ptts_request* req, ptts_result* res, std::string* err) {
    std::memset(res, 0, sizeof *res);
    res->eos_step = -1;
    DispatchItem item;
    item.req = req; item.res = res;
    if (!d->models.empty()) {
        std::string e = request_error(d->models[0]->d, *req);   // refuse malformed requests before they cost a batch slot
        if (!e.empty()) { *err = e; res->status = PTTS_EINVAL; return PTTS_EINVAL; }
        if (req->voice) {
            item.voice = reinterpret_cast<const Voice*>(req->voice);
            bool served = false;
            for (Model* m : d->models) served |= voice_usable_by(*item.voice, *m);
            if (!served) { *err = "ptts-hip: voice belongs to a model this dispatcher does not serve"; res->status = PTTS_EINVAL; return PTTS_EINVAL; }
        }
    }
    std::unique_lock<std::mutex> lock(d->mu);
    if (d->closing) { *err = "dispatcher closed"; res->status = PTTS_ECANCELLED; return PTTS_ECANCELLED; }
    if ((int)d->queue.size() >= d->queue_cap) { *err = "dispatcher: queue full"; res->status = PTTS_ENOMEM; return PTTS_ENOMEM; }
    item.enq = Clock::now();
    d->last_arrival = item.enq;
    d->queue.push_back(&item);
    d->max_depth = std::max<int64_t>(d->max_depth, (int64_t)d->queue.size());
    d->cv_work.notify_all();
    item.cv.wait(lock, [&] { return item.done; });
    if (item.rc != PTTS_OK) *err = item.err;
    return item.rc;
}

void dispatcher_stats(Dispatcher* d, ptts_dispatch_stats* out) {
    std::lock_guard<std::mutex> lock(d->mu);
    std::memset(out, 0, sizeof *out);
    out->requests = d->n_requests;
    out->batches = d->n_batches;
    out->cancelled_waiting = d->n_cancelled_waiting;
    out->max_queue_depth = d->max_depth;
    out->mean_batch = d->n_batches ? (double)d->n_requests / (double)d->n_batches : 0.0;
    out->mean_wait_us = d->n_requests ? d->sum_wait_us / (double)d->n_requests : 0.0;
    out->cont_steps = d->cont_steps; out->cont_slot_steps = d->cont_slot_steps;
    out->mean_exec_us = d->n_batches ? d->sum_exec_us / (double)d->n_batches : 0.0;
    for (Model* m : d->models) out->flow_cluster_fallbacks += m->fc_fallbacks.load();
}

}  // namespace ptts
