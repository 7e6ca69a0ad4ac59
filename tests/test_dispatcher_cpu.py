"""Queueing rules of the request dispatcher over a Python executor: no GPU involved.

The reference admits `workers` concurrent Synthesize calls through a semaphore and lets the others wait, answering 503 to a
request cancelled while waiting (internal/server/server.go:132-134,398-421).  Here waiting requests are coalesced into
batches; these tests pin the coalescing window, the batch cap, arrival order, cancellation-while-waiting and error
propagation."""
import threading
import time

import numpy as np
import pytest


def run_clients(disp, pkg, n, cfgs=None, stagger=0.0):
    out, errs = [None] * n, [None] * n

    def client(i):
        try:
            cfg = cfgs[i] if cfgs else pkg.RuntimeGenerateConfig()
            out[i] = disp.generate([i + 1, 7, 9], cfg)
        except Exception as e:  # noqa: BLE001
            errs[i] = e

    ts = []
    for i in range(n):
        t = threading.Thread(target=client, args=(i,))
        t.start()
        ts.append(t)
        if stagger:
            time.sleep(stagger)
    for t in ts:
        t.join(30)
        assert not t.is_alive()
    return out, errs


def make_exec(batches, delay=0.0, fail=None):
    def ex(_user, _worker, reqs, n, results, err, errlen):
        batches.append([int(reqs[i].tokens[0]) for i in range(n)])
        if delay:
            time.sleep(delay)
        for i in range(n):
            results[i].n_frames = int(reqs[i].tokens[0])   # echo: every caller must get ITS result back
            results[i].status = 0
        if fail:
            import ctypes
            ctypes.memmove(err, fail.encode() + b"\0", min(errlen, len(fail) + 1))
            return 1
        return 0
    return ex


def test_waiting_requests_are_coalesced_up_to_max_batch(pkg):
    batches = []
    d = pkg.Dispatcher([], max_batch=8, window_us=150_000, _custom_exec=make_exec(batches, delay=0.05))
    out, errs = run_clients(d, pkg, 20)
    assert not any(errs)
    assert sorted(o.n_frames for o in out) == list(range(1, 21))          # each caller got its own result
    assert [o.n_frames for o in out] == list(range(1, 21))
    assert all(len(b) <= 8 for b in batches) and sum(len(b) for b in batches) == 20
    st = d.stats()
    assert st["requests"] == 20 and st["batches"] == len(batches) and st["mean_batch"] > 2.0
    d.close()


def test_lone_request_waits_one_window_and_a_full_batch_leaves_at_once(pkg):
    batches = []
    d = pkg.Dispatcher([], max_batch=4, window_us=200_000, _custom_exec=make_exec(batches))
    t0 = time.perf_counter()
    run_clients(d, pkg, 1)
    lone = time.perf_counter() - t0
    assert 0.18 <= lone < 1.0, lone
    d.close()
    d = pkg.Dispatcher([], max_batch=4, window_us=5_000_000, _custom_exec=make_exec(batches))
    t0 = time.perf_counter()
    run_clients(d, pkg, 4)
    assert time.perf_counter() - t0 < 2.0          # did not sit out the 5 s window
    assert sorted(batches[-1]) == [1, 2, 3, 4]
    d.close()


def test_arrival_order_is_kept(pkg):
    batches = []
    d = pkg.Dispatcher([], max_batch=3, window_us=400_000, _custom_exec=make_exec(batches))
    run_clients(d, pkg, 6, stagger=0.02)
    assert [x for b in batches for x in b] == [1, 2, 3, 4, 5, 6]
    d.close()


def test_request_cancelled_while_waiting_never_runs(pkg):
    batches = []
    d = pkg.Dispatcher([], max_batch=1, window_us=0, _custom_exec=make_exec(batches, delay=0.4))
    flag = np.zeros(1, np.int32)
    cfgs = [pkg.RuntimeGenerateConfig(), pkg.RuntimeGenerateConfig(cancel=flag)]
    res = {}

    def late():
        time.sleep(0.1)      # request 1 is running (one worker, batch of 1), request 2 waits in the queue
        flag[0] = 1

    threading.Thread(target=late).start()
    out, errs = run_clients(d, pkg, 2, cfgs=cfgs, stagger=0.03)
    assert errs[0] is None and out[0].n_frames == 1
    assert isinstance(errs[1], pkg.Cancelled) and "cancelled while waiting" in str(errs[1])
    assert batches == [[1]]
    assert d.stats()["cancelled_waiting"] == 1
    d.close()
    del res


def test_executor_errors_reach_every_caller_of_the_batch(pkg):
    d = pkg.Dispatcher([], max_batch=4, window_us=100_000, _custom_exec=make_exec([], fail="generate: boom"))
    out, errs = run_clients(d, pkg, 3)
    assert all(isinstance(e, pkg.PttsError) and "boom" in str(e) for e in errs)
    d.close()


def test_two_workers_drain_one_queue(pkg):
    seen = []
    lock = threading.Lock()

    def ex(_u, worker, reqs, n, results, err, errlen):
        with lock:
            seen.append(worker)
        time.sleep(0.15)
        for i in range(n):
            results[i].n_frames = int(reqs[i].tokens[0])
        return 0

    d = pkg.Dispatcher([], max_batch=2, window_us=1000, _custom_exec=ex, _workers=2)
    t0 = time.perf_counter()
    out, errs = run_clients(d, pkg, 8)
    dt = time.perf_counter() - t0
    assert not any(errs) and sorted(o.n_frames for o in out) == list(range(1, 9))
    assert set(seen) == {0, 1}               # both workers took batches ...
    assert dt < 0.15 * 4 * 0.9 + 0.3         # ... concurrently: 4 batches of 2 in about two rounds, not four
    d.close()


def test_window_stretches_while_requests_keep_arriving(pkg):
    """A burst that trickles in (one request every 8 ms, window 50 ms, so window / 4 = 12.5 ms counts as quiet) is not cut at
    the window: the batch leaves once the arrivals pause, at most four windows after the first request."""
    batches = []
    d = pkg.Dispatcher([], max_batch=64, window_us=50_000, _custom_exec=make_exec(batches))
    t0 = time.perf_counter()
    run_clients(d, pkg, 12, stagger=0.008)         # arrivals over ~90 ms: past the 50 ms window
    dt = time.perf_counter() - t0
    assert batches == [list(range(1, 13))], batches
    assert dt < 0.2 + 0.1                          # hard stop at 4 windows
    batches.clear()
    run_clients(d, pkg, 40, stagger=0.008)         # ~320 ms of arrivals: the 200 ms cap cuts the burst
    assert len(batches) >= 2 and sum(len(b) for b in batches) == 40
    d.close()


def test_two_workers_do_not_split_a_trickle(pkg):
    """Two workers finish together and find six requests that have waited longer than a window: the first takes a full
    batch, and the two it leaves behind get a FRESH window (late arrivals join them) instead of leaving at once as a sliver."""
    batches = []
    d = pkg.Dispatcher([], max_batch=4, window_us=100_000, _custom_exec=make_exec(batches, delay=0.2), _workers=2)
    res = {}

    def wave(n, at):
        time.sleep(at)
        res[at] = run_clients(d, pkg, n)

    ts = [threading.Thread(target=wave, args=a) for a in ((8, 0.0), (6, 0.05), (2, 0.23))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    assert all(not any(errs) for _, errs in res.values())
    assert sorted(len(b) for b in batches) == [4, 4, 4, 4], batches
    d.close()
