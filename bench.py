#!/usr/bin/env python3
"""Headline benchmark: synthesized audio seconds per wall second (xRT) of the MI355X PocketTTS path.

One "step" = one pass of the hot path over one batch of synthetic utterance chunks:
tts.Runtime.GenerateAudio for 64 concurrent utterances per GPU (text prompt of 25 tokens on top of a
125-frame voice model state, prefill, 125 autoregressive steps = 10 s of audio each, Mimi decode to
24 kHz PCM).  BASELINE.json configs[2] at N=1 and configs[3] (64 per GPU) at N=8: weak scaling, no
data-path collective -- one RCCL broadcast of the weight arena at start-up.

    python bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches its N ranks itself (one child process per
GPU, before anything in the parent touches the GPU); under `python -m torch.distributed.run ... bench.py --gpus N` it is
one of the ranks.  Rank 0 prints one JSON line.  `roofline` prices the dominant kernel (the AR step's weight-streaming
linear) against HBM, `roofline.mimi` the decoder against the bf16 matrix peak; `cpu_baseline` times the CPU oracle (a C
restatement of the reference's AVX2/FMA path: the Go binary itself cannot be built here) on the host cores, on a bounded
sample, on rank 0 at N=1 only.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
FRAME_SEC = 0.08           # 1920 samples @ 24 kHz (PLAN.md:37)
# algorithmic work of the decoder per 80 ms frame per utterance
MIMI_FLOP_PER_FRAME = 2 * 270.7e6
MIMI_ACT_BYTES_PER_FRAME = 2.2e6   # activations, every layer of the SEANet ladder crossing HBM once (f32)

WORKLOADS = {
    # name: (batch per GPU, frames, file dtype, weights mode, kv mode, dtype label)
    "b64_10s_bf16": dict(batch=64, frames=125, file="BF16", weights=1, kv=1, dtype="bf16"),
    "b1_5s_f32": dict(batch=1, frames=63, file="F32", weights=0, kv=0, dtype="f32"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def checkpoint_path(pkg, file_dtype: str, rank: int, barrier) -> str:
    d = os.path.join(tempfile.gettempdir(), f"ptts_bench_{os.getuid()}")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, f"synthetic_b6369a24_shapes_{file_dtype.lower()}.safetensors")
    if rank == 0 and not os.path.exists(path):
        t0 = time.time()
        tensors = pkg.synth.make_checkpoint(pkg.synth.SynthConfig.full(), seed=1234)
        tmp = path + f".tmp{os.getpid()}"
        pkg.synth.write_safetensors(tmp, tensors, dtype=file_dtype)
        os.replace(tmp, path)
        log(f"[bench] synthetic checkpoint ({file_dtype}) written in {time.time()-t0:.1f}s: {path}")
    barrier()
    return path


def voice_modules(pkg, cfg):
    tens = pkg.synth.make_voice_state(cfg, offset=125, seed=7)
    mods = {}
    for name, t in tens.items():
        mod, key = name.rsplit("/", 1)
        mods.setdefault(mod, {})[key] = np.asarray(t, np.float32)
    return mods


def shard_prompts(all_prompts, rank: int, per_rank: int):
    """Utterance chunks are independent (service.go:138-153): rank r owns rows [r*per_rank, (r+1)*per_rank). No exchange."""
    return all_prompts[rank * per_rank:(rank + 1) * per_rank]


def max_over_ranks(value: float, world: int, device=None) -> float:
    if world == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value: float, world: int, device=None) -> list:
    if world == 1:
        return [value]
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


# ------------------------------------------------------------------------------------------------------------------
# rank launch: `python bench.py --gpus N` with no rendezvous in the environment starts its own N ranks
# ------------------------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv: list) -> int:
    """One child per GPU (RANK = LOCAL_RANK = i), rendezvous on 127.0.0.1.  The parent makes no GPU / HIP call -- it only
    waits -- and no process is ever replaced (fresh children, not exec of a process that has touched the GPU).  The worker
    pool this mirrors: internal/server/server.go:119-143,398-421 (N workers behind one front door)."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), PTTS_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                c = procs[r].poll()
                if c is None:
                    continue
                pending.discard(r)
                if c != 0:
                    rc = rc or c
                    log(f"[bench] rank {r} exited with code {c}; stopping the other ranks")
                    for q in pending:
                        procs[q].terminate()   # the exact children started above
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


NATIVE_BROADCAST = False   # --native-broadcast: the library's own ncclBroadcast (ptts_rccl_broadcast) instead of torch.distributed's


def open_model(pkg, path, wl, rank, world, device):
    """Two-phase open: rank 0 fills the device arena, one RCCL broadcast hands it to the other ranks."""
    kw = dict(device=device, weights=wl["weights"], kv=wl["kv"], max_batch=wl["batch"], use_graph=True)   # configs[2]: hipGraph-captured step
    if world == 1:
        return pkg.Model.open(path, **kw), None
    import torch
    import torch.distributed as dist
    plan, nbytes = pkg.Model.plan(path, **kw)
    arena = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{device}")
    if rank == 0:
        model = pkg.Model.open_planned(plan, arena.data_ptr(), fill=True)
        torch.cuda.synchronize()
    if NATIVE_BROADCAST:                  # the same broadcast issued by libptts_hip.so itself (what a host without PyTorch calls)
        ids = [pkg.runtime.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0, device=torch.device(f"cuda:{device}"))
        pkg.runtime.rccl_broadcast(arena.data_ptr(), nbytes, rank, world, ids[0], device)
    else:
        dist.broadcast(arena, src=0)      # the only collective of the whole job (weights, once)
    torch.cuda.synchronize()
    # every rank holds the same bytes (checked, not assumed): a 64-bit sum of the arena, compared across ranks
    pad = (-nbytes) % 8
    cs = torch.sum(torch.nn.functional.pad(arena, (0, pad)).view(torch.int64)) if pad else torch.sum(arena.view(torch.int64))
    sums = [torch.zeros_like(cs) for _ in range(world)]
    dist.all_gather(sums, cs)
    if len({int(x.item()) for x in sums}) != 1:
        raise SystemExit(f"bench.py (rank {rank}): the weight arena differs between ranks after the broadcast")
    if rank != 0:
        model = pkg.Model.open_planned(plan, arena.data_ptr(), fill=False)
    return model, arena


STEP_WEIGHT_PARAMS = 85_263_360     # parameters every AR step streams once for the whole batch
KV_ELEMS_PER_KEY = 2 * 6 * 1024      # K and V, 6 layers, d_model 1024: elements one cached position holds per utterance
PROMPT_KEYS = 125 + 25               # the benchmark's cache before the first step: 125-frame voice state + 25 prompt tokens


def step_level(batch, frames, ar_loop_ms, weight_elem=2, kv_elem=2):
    """The WHOLE AR step against HBM (beside the per-kernel entry): SURVEY.md 8(d)'s bytes per step -- every step weight once for the batch
    plus the keys and values each utterance's attention reads (cache length averaged over the run) -- over the measured time of a step."""
    us = 1e3 * ar_loop_ms / max(1, frames)
    mean_keys = PROMPT_KEYS + (frames - 1) / 2.0
    nbytes = STEP_WEIGHT_PARAMS * weight_elem + batch * KV_ELEMS_PER_KEY * kv_elem * mean_keys
    gbs = nbytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "us_per_step": round(us, 2),
            "algorithmic_bytes_per_step": round(nbytes), "rows": batch,
            "note": "step-level: (step weights once + KV bytes read by the batch, SURVEY.md 8d) / (AR-loop device time / steps)"}


def step_level_fraction(batch, frames, step_us, weight_elem=2, kv_elem=2):
    return step_level(batch, frames, step_us * frames * 1e-3, weight_elem, kv_elem)["frac"]


def wide_batch_pass(pkg, model, wl, batch, voice, steps, sync):
    """Labelled extra (never the headline): the same 10-s utterances, `batch` of them per call through ONE engine (the AR step's kernels take up to 256
    rows: row tiles of the step linear, 12-row tiles of the flow-net cluster; the decoder goes through its workspace in groups of 64)."""
    eng = model.share()
    try:
        eng.set_max_batch(batch)
        eng.set_use_graph(True)
        wlb = dict(wl, batch=batch)
        prompts = pkg.synth.make_prompts(batch, 25, 4000, seed=42)
        e, lat, _ = run_workload(pkg, eng, wlb, prompts, voice, steps, 2, lambda: None, sync)
        cfgs = gen_cfgs(pkg, wlb, batch, voice)
        toks = [np.ascontiguousarray(p, np.int64) for p in prompts]
        eng.profile_enable(2)
        try:
            eng.generate_batch(toks, cfgs)
            ph = eng.profile_read()
        finally:
            eng.profile_enable(False)
        return {"value": round(batch * wl["frames"] * FRAME_SEC * steps / e, 1), "unit": "x real-time", "ms_per_step": round(1e3 * e / steps, 3),
                "p50_utterance_latency_ms": round(1e3 * statistics.median(lat), 2),
                "phases_ms": {"prefill": round(ph["prefill_ms"], 3), "ar_loop": round(ph["ar_loop_ms"], 3), "mimi": round(ph["mimi_ms"], 3)},
                "step_level": step_level(batch, wl["frames"], ph["ar_loop_ms"]),
                "config": f"labelled extra, not the headline: {batch} utterances x {wl['frames']} frames in ONE call of one engine (ptts_model_set_max_batch {batch}), "
                          "graph replay, bf16 weights + KV, the decoder in groups of 64 utterances"}
    finally:
        eng.close()


def gen_cfgs(pkg, wl, n, voice, **kw):
    base = dict(temperature=0.0, eos_threshold=float("inf"), max_steps=wl["frames"], lsd_decode_steps=1, frames_after_eos=3, device_voice=voice)
    base.update(kw)
    return [pkg.RuntimeGenerateConfig(**base) for _ in range(n)]


def run_client_threads(client, clients, n):
    """clients closed-loop client threads, each client(i, n); every thread that started is joined whatever happens (a rank that cannot start them all -- a
    task limit on a node that carries eight ranks -- reports the failure instead of leaving requests in flight behind its back)."""
    import threading
    failures, started = [], []

    def guarded(i):
        try:
            client(i, n)
        except Exception as e:  # noqa: BLE001
            failures.append(e)

    try:
        for i in range(clients):
            t = threading.Thread(target=guarded, args=(i,))
            t.start()
            started.append(t)
    except RuntimeError as e:      # "can't start new thread"
        failures.append(e)
    for t in started:
        t.join()
    if failures:
        raise failures[0]


def two_barrier_pass(barrier, sync, prepare, timed):
    """prepare() builds and warms (untimed), timed() is the measured round; returns (wall seconds of timed(), the exception or None).  EVERY rank reaches both
    barriers whatever happens on it -- a rank whose engine failed must not leave the others waiting inside a collective."""
    err, dt = None, 0.0
    try:
        prepare()
    except Exception as e:  # noqa: BLE001
        err = e
    sync(); barrier()
    if err is None:
        try:
            t0 = time.perf_counter()
            timed()
            sync()
            dt = time.perf_counter() - t0
        except Exception as e:  # noqa: BLE001
            err = e
    barrier()
    return dt, err


def serve_pass(pkg, model, wl, prompts, voice, barrier, sync, clients=128, per_client=4):
    """BASELINE.json configs[3] names the serve-mode worker pool (internal/server/server.go:119-143,398-421): this rank's share of it is one
    dispatcher over two engines of its GPU (one weight arena), fed by closed-loop clients that each synthesise 10-s utterances back to back
    (64 per engine in flight).  Returns (audio seconds, wall seconds, p50 latency) of the timed rounds; an untimed round comes first."""
    import threading
    st = {"m2": None, "disp": None}
    lat = []
    lock = threading.Lock()
    cfg = gen_cfgs(pkg, wl, 1, voice, pcm16=True)[0]
    toks = [p.tolist() for p in prompts]

    def client(i, n):
        for k in range(n):
            t0 = time.perf_counter()
            r = st["disp"].generate(toks[(i + k) % len(toks)], cfg)
            with lock:
                lat.append(time.perf_counter() - t0)
            assert r.n_frames == wl["frames"]

    def round_(n):
        run_client_threads(client, clients, n)

    def prepare():
        st["m2"] = m2 = model.share()
        for m in (model, m2):
            m.set_use_graph(False)   # plain launches: the dispatcher's default, and no idle gap between replays when two engines interleave
        # every engine once at full batch before anything is timed: an engine's first call allocates its KV caches, ~16 GB of decoder workspace and its share of
        # the page-locked result pool (~0.5 s) -- the untimed round below does not always reach both engines (its two batches may land on the same one), and an
        # engine first used inside the timed rounds then costs half a second of them (seen once: 2.3 k instead of 15.9 k x real time)
        for m in (model, m2):
            m.generate_batch(toks[:wl["batch"]], [cfg] * min(wl["batch"], len(toks)))
        st["disp"] = pkg.Dispatcher([model, m2], max_batch=wl["batch"], window_us=3000)
        round_(1)
        lat.clear()

    try:
        dt, err = two_barrier_pass(barrier, sync, prepare, lambda: round_(per_client))
        if err is not None:
            raise err
        mean_batch = st["disp"].stats()["mean_batch"]
    finally:
        if st["disp"] is not None:
            st["disp"].close()
        if st["m2"] is not None:
            st["m2"].close()
        model.set_use_graph(True)
    return clients * per_client * wl["frames"] * FRAME_SEC, dt, statistics.median(lat), mean_batch


def serve_continuous_pass(pkg, model, wl, voice, barrier, sync, slots=192, clients=384, per_client=4, mixed=True):
    """The serving legs through the dispatcher's DEFAULT configuration for one model on its GPU: ONE continuous-batching engine of `slots` utterances
    (csrc/continuous.cpp: slots refilled between groups of AR steps, finished utterances decoded beside -- or, a wave of them, between -- the following steps;
    internal/server/server.go:398-421 + internal/tts/service.go:138-153 are the reference's counterparts), closed-loop clients.  mixed: what real traffic looks
    like once EOS is finite -- utterances of 2-12 s (budgets drawn uniformly from 25..150 frames per request); else every request is the workload's 10 s.
    Returns (audio seconds, wall seconds, p50 latency, mean occupied slots, flow-cluster fallbacks) of the timed round."""
    import random
    import threading
    st = {"eng": None, "disp": None, "st0": None}
    prompts = [p.tolist() for p in pkg.synth.make_prompts(256, 25, 4000, seed=3)]
    plan = random.Random(5)
    frames = [plan.randint(25, 150) if mixed else wl["frames"] for _ in range(4096)]
    lat, done = [], []
    lock = threading.Lock()

    def client(i, n):
        for k in range(n):
            nf = frames[(i * per_client + k) % len(frames)]
            c = pkg.RuntimeGenerateConfig(temperature=0.0, eos_threshold=float("inf"), max_steps=nf, lsd_decode_steps=1, frames_after_eos=3, device_voice=voice, pcm16=True)
            t0 = time.perf_counter()
            r = st["disp"].generate(prompts[(i * per_client + k) % len(prompts)], c)
            dt = time.perf_counter() - t0
            assert r.n_frames == nf
            with lock:
                lat.append(dt)
                done.append(nf)

    def round_(n):
        run_client_threads(client, clients, n)

    def prepare():
        st["eng"] = eng = model.share()
        eng.set_max_batch(slots)
        eng.set_use_graph(False)
        warm = gen_cfgs(pkg, wl, 1, voice, pcm16=True)[0]
        eng.generate_batch(prompts[:slots], [warm] * slots)     # the engine's buffers (KV caches at this width, result pool) exist before anything is timed
        st["disp"] = pkg.Dispatcher([eng], max_batch=slots, window_us=3000)   # the default: continuous batching (one model alone on its GPU), 512 keys, 256 steps
        round_(1)
        lat.clear(); done.clear()
        st["st0"] = st["disp"].stats()

    try:
        dt, err = two_barrier_pass(barrier, sync, prepare, lambda: round_(per_client))
        if err is not None:
            raise err
        s1, s0 = st["disp"].stats(), st["st0"]
        occ = (s1["cont_slot_steps"] - s0["cont_slot_steps"]) / max(1, s1["cont_steps"] - s0["cont_steps"])
        return sum(done) * FRAME_SEC, dt, statistics.median(lat), occ, s1["flow_cluster_fallbacks"]
    finally:
        if st["disp"] is not None:
            st["disp"].close()
        if st["eng"] is not None:
            st["eng"].close()


def run_workload(pkg, model, wl, prompts, voice, steps, warmup, barrier, sync, **cfg_kw):
    cfgs = gen_cfgs(pkg, wl, len(prompts), voice, **cfg_kw)
    toks = [np.ascontiguousarray(p, np.int64) for p in prompts]
    out = None
    for _ in range(warmup):
        out = model.generate_batch(toks, cfgs)   # held like in the timed loop: the pinned result pool reaches its steady state (two sets)
    # (the interpreter's cyclic collector stays out of the timed region: a full collection of this process -- torch and numpy imported -- is a 40-50 ms pause, and it
    # came at the same step of every 128-utterance run; nothing the loop allocates is cyclic)
    import gc
    gc.collect()
    gc.disable()
    try:
        sync(); barrier()
        lat = []
        t0 = time.perf_counter()
        for _ in range(steps):
            s0 = time.perf_counter()
            out = model.generate_batch(toks, cfgs)
            lat.append(time.perf_counter() - s0)
        sync(); barrier()
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()
    log("[bench] per-step ms: " + " ".join(f"{1e3*x:.1f}" for x in lat))
    frames = sum(o.n_frames for o in out)
    assert all(o.n_frames == wl["frames"] and o.pcm.shape[0] == wl["frames"] * 1920 for o in out)
    assert os.environ.get("PTTS_PROBE_GARBAGE") or all(np.isfinite(o.pcm).all() for o in out[:2])   # (ablation builds of tools/probes feed garbage on purpose: timing only)
    return elapsed, lat, frames


def roofline_pass(pkg, model, wl, prompts, voice, traffic=None):
    """One extra pass with HIP events around every launch of the dominant kernel (eager, same stream) and around the phases.
    `achieved` counts the ALGORITHMIC bytes of SURVEY.md 8(d) that go through this kernel -- every step weight once per launch
    (85.26 M params: 170.5 MB bf16 / 341 MB f32 per AR step, shared by the whole batch) -- over the sum of the launch
    durations; the same with the activation rows in and the outputs counted is reported beside it."""
    cfgs = gen_cfgs(pkg, wl, len(prompts), voice)
    toks = [np.ascontiguousarray(p, np.int64) for p in prompts]
    model.generate_batch(toks, cfgs)   # untimed: whatever ran last on this model (another temperature, plain launches) may have dropped its step graphs -- re-capturing them belongs to no phase
    model.profile_enable(2)   # phases only: the call runs as configured (graph replay), HIP events at the phase boundaries
    try:
        model.generate_batch(toks, cfgs)
        phases = model.profile_read()
    finally:
        model.profile_enable(False)
    model.profile_enable(True)   # every launch of the dominant kernel timed (plain launches)
    try:
        model.generate_batch(toks, cfgs)
        prof = model.profile_read()
    finally:
        model.profile_enable(False)
    for k in ("prefill_ms", "ar_loop_ms", "mimi_ms"):
        prof[k] = phases[k]
    sec = prof["total_ms"] * 1e-3
    n = max(1, prof["launches"])
    achieved = prof["weight_bytes"] / sec / 1e9 if sec > 0 else 0.0
    with_act = prof["algorithmic_bytes"] / sec / 1e9 if sec > 0 else 0.0
    r = {"bound": "hbm", "kernel": prof["kernel"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "launches": prof["launches"],
         "avg_launch_us": round(prof["total_ms"] * 1e3 / n, 3),
         "algorithmic_bytes_per_launch": round(prof["weight_bytes"] / n),
         "algorithmic_bytes_note": "weights only: each step linear's matrix once per launch (SURVEY.md 8d: 85.26 M params per AR step)",
         "achieved_with_activations": round(with_act, 1), "frac_with_activations": round(with_act / HBM_PEAK_GBS, 4),
         "bytes_per_launch_with_activations": round(prof["algorithmic_bytes"] / n),
         "phases_ms": {"prefill": round(prof["prefill_ms"], 3), "ar_loop": round(prof["ar_loop_ms"], 3), "mimi": round(prof["mimi_ms"], 3)}}
    r["step_level"] = step_level(len(prompts), wl["frames"], prof["ar_loop_ms"], {0: 4, 1: 2, 2: 1}[wl["weights"]], 2 if wl["kv"] else 4)
    # the same kernel's mean as rocprofv3 reports it for the same workload under graph replay (a child pass of this run).  Under the profiler every
    # launch is a little longer and the chip's clock a little lower (2-3 %), the in-process events see the kernels launched one by one; where the two
    # differ, `frac` is priced with the LONGER one, and both stand in the line.
    tr = (traffic or {}).pop("_trace", None) if isinstance(traffic, dict) else None
    if tr and tr[0]:
        r["avg_launch_us_rocprof"] = round(tr[0], 3)
        r["launches_rocprof"] = tr[1]
        slow = max(tr[0], r["avg_launch_us"])
        r["avg_launch_us_events"] = r["avg_launch_us"]
        r["achieved_events"] = r["achieved"]
        r["achieved"] = round(r["algorithmic_bytes_per_launch"] / (slow * 1e-6) / 1e9, 1)
        r["frac"] = round(r["achieved"] / HBM_PEAK_GBS, 4)
        r["frac_note"] = "achieved / frac use the longer of the two mean launch durations (in-process events, rocprofv3 kernel trace of the same workload)"
    elif tr:
        r["avg_launch_us_rocprof_note"] = tr[2]
    if tr and len(tr) > 3 and tr[3]:
        r["flow_cluster"] = {"avg_launch_us_rocprof": round(tr[3][0], 3), "launches_rocprof": tr[3][1],
                             "note": "k_flow_cluster: the flow net's 12 residual-block linears (6.3 MB of the step's 170.5 MB of weights) as one launch per AR step; not a launch of the dominant kernel, not in avg_launch_us"}
    if traffic and traffic.get(prof["kernel"]):
        t = traffic[prof["kernel"]]
        r["traffic"] = t.get("bytes_per_launch")
        r["traffic_detail"] = t
    elif traffic is not None:
        r["traffic_detail"] = traffic
    # the decoder against the matrix peak: algorithmic FLOP (270.7 M MAC per frame per utterance) over its device time
    if prof["mimi_ms"] > 0:
        frames = len(prompts) * wl["frames"]
        flop = frames * MIMI_FLOP_PER_FRAME
        tf = flop / (prof["mimi_ms"] * 1e-3) / 1e12
        gbs = frames * MIMI_ACT_BYTES_PER_FRAME / (prof["mimi_ms"] * 1e-3) / 1e9
        r["mimi"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                     "ms": round(prof["mimi_ms"], 3), "algorithmic_flop": flop,
                     "note": "algorithmic FLOP; every product runs as 2 (bf16 weights) or 3 (f32 weights) bf16 MFMAs on hi/lo-split f32 activations",
                     "seanet_activation_bytes": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}}
    return r


# ------------------------------------------------------------------------------------------------------------------
# HBM traffic of the dominant kernel from the PMC counters, measured by this run: two rocprofv3 passes (reads, writes) over a
# short batch-64 probe, as child processes, before this process touches the GPU.  MI355X_MICROARCH.md (HBM section): FETCH_SIZE
# (KiB) reports half of the bytes of wide coalesced reads on gfx950 -> doubled; WRITE_SIZE (KiB) is exact for 16-byte stores.
# ------------------------------------------------------------------------------------------------------------------
def _pmc_pass(counters, out_dir, steps):
    exe = shutil.which("rocprofv3") or next((p for p in ("/opt/rocm/bin/rocprofv3",) if os.path.exists(p)), None)
    if not exe:
        return None, "rocprofv3 not found"
    env = dict(os.environ, PTTS_PROBE_STEPS=str(steps), TMPDIR=tempfile.gettempdir())
    cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", out_dir, "-o", "pmc", "--", sys.executable,
                                                           os.path.join(ROOT, "tools", "traffic_probe.py")]
    try:
        p = subprocess.run(cmd, env=env, cwd=tempfile.gettempdir(), capture_output=True, text=True, timeout=420)
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 pass timed out"
    files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
    if p.returncode != 0 or not files:
        return None, f"rocprofv3 {' '.join(counters)} failed (rc {p.returncode}): {(p.stderr or '')[-300:]}"
    per = {}
    for r in csv.DictReader(open(files[0])):
        base = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].split("::")[-1].strip()
        d = per.setdefault(base, {})
        v = d.setdefault(r["Counter_Name"], [0.0, 0])
        v[0] += float(r["Counter_Value"])
        v[1] += 1
    return per, None


def trace_pass(steps, kernel="k_skinny"):
    """rocprofv3 --kernel-trace over the headline workload (tools/traffic_probe.py with graph replay and `steps` frames per utterance, a child
    process): the launch-weighted mean duration of the dominant kernel as the profiler itself reports it -- the figure profiles/r4_by_grid.txt
    holds -- to stand in the bench line beside the in-process (hipExtLaunchKernel event) figure.  Returns (mean_us, launches, note)."""
    exe = shutil.which("rocprofv3") or next((p for p in ("/opt/rocm/bin/rocprofv3",) if os.path.exists(p)), None)
    if not exe:
        return None, 0, "rocprofv3 not found"
    out_dir = tempfile.mkdtemp(prefix="ptts_trace_")
    try:
        env = dict(os.environ, PTTS_PROBE_STEPS=str(steps), PTTS_PROBE_GRAPH="1", PTTS_PROBE_REPS="2", TMPDIR=tempfile.gettempdir())
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "tr", "--", sys.executable, os.path.join(ROOT, "tools", "traffic_probe.py")]
        try:
            p = subprocess.run(cmd, env=env, cwd=tempfile.gettempdir(), capture_output=True, text=True, timeout=420)
        except subprocess.TimeoutExpired:
            return None, 0, "rocprofv3 kernel-trace pass timed out"
        files = glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True)
        if p.returncode != 0 or not files:
            return None, 0, f"rocprofv3 --kernel-trace failed (rc {p.returncode}): {(p.stderr or '')[-300:]}"
        tot, n, fc_tot, fc_n = 0, 0, 0, 0
        for r in csv.DictReader(open(files[0])):
            if kernel in r["Kernel_Name"]:
                tot += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                n += 1
            elif "k_flow_cluster" in r["Kernel_Name"]:   # the flow net's residual blocks: one launch per AR step beside the step linears
                fc_tot += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                fc_n += 1
        return (tot / n / 1e3 if n else None), n, None, ((fc_tot / fc_n / 1e3, fc_n) if fc_n else None)
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def measure_traffic(steps=12):
    out = {}
    notes = []
    tmp = tempfile.mkdtemp(prefix="ptts_pmc_")
    try:
        rd, err = _pmc_pass(["FETCH_SIZE"], os.path.join(tmp, "rd"), steps)
        if err:
            notes.append(err)
        wr, err = _pmc_pass(["WRITE_SIZE"], os.path.join(tmp, "wr"), steps)
        if err:
            notes.append(err)
            wr, err2 = _pmc_pass(["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"], os.path.join(tmp, "wr2"), steps)
            if err2:
                notes.append(err2)
        for k in set(rd or {}) | set(wr or {}):
            e = {}
            if rd and "FETCH_SIZE" in rd.get(k, {}):
                s, n = rd[k]["FETCH_SIZE"]
                e["read_bytes_per_launch"] = round(s * 1024.0 * 2.0 / n)
                e["launches"] = n
            if wr and "WRITE_SIZE" in wr.get(k, {}):
                s, n = wr[k]["WRITE_SIZE"]
                e["write_bytes_per_launch"] = round(s * 1024.0 / n)
            elif wr and "TCC_EA0_WRREQ_sum" in wr.get(k, {}):
                s, n = wr[k]["TCC_EA0_WRREQ_sum"]
                s64 = wr[k].get("TCC_EA0_WRREQ_64B_sum", [0.0, n])[0]
                e["write_bytes_per_launch"] = round((s64 * 64.0 + (s - s64) * 32.0) / n)   # 64-byte requests + the rest counted as 32-byte
            if "read_bytes_per_launch" in e and "write_bytes_per_launch" in e:
                e["bytes_per_launch"] = e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
            out[k] = e
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["_source"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run over tools/traffic_probe.py (batch 64, bf16, {steps} AR steps); "
                      "FETCH_SIZE KiB x 1024 x 2 (gfx950 correction), WRITE_SIZE KiB x 1024; mean per launch")
    if notes:
        out["_notes"] = notes
    return out


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle at the reference's default 2 workers and at all physical cores, batch 1,
# 1 warm-up + >= 5 timed utterances each, p50
# ------------------------------------------------------------------------------------------------------------------
def host_cpu():
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    physical = min(len(cores) or logical, usable)
    return model, logical, usable, physical


def cpu_baseline(pkg, path, cfg, frames=63, runs=5):
    from oracle import oracle as O
    O.build()
    om = O.OracleModel.from_file(path)
    mods = voice_modules(pkg, cfg)
    toks = pkg.synth.make_prompts(1, 25, 4000, seed=42)[0]
    model, logical, usable, physical = host_cpu()
    O.set_use_avx2(True)

    def one(workers, n_frames):
        O.set_workers(workers, workers)   # conv-workers / runtime-workers (config.go:76-83, service.go:318-328)
        t0 = time.perf_counter()
        r = om.generate(toks, max_steps=n_frames, eos_threshold=1e30, frames_after_eos=3, voice_state=mods)
        assert r["n_frames"] == n_frames
        return time.perf_counter() - t0

    def leg(workers):   # 1 warm-up + >= 5 timed utterances, p50
        one(workers, frames)
        times = [one(workers, frames) for _ in range(runs)]
        p50 = statistics.median(times)
        return {"workers": workers, "p50_latency_ms": round(1e3 * p50, 1), "xrt": round(frames * FRAME_SEC / p50, 3), "timed_runs": runs, "frames": frames}

    legs = {"reference_default_2_workers": leg(2)}
    # the reference's fork-join split (one goroutine per worker and per operator) stops paying long before a 128-core host is
    # full: a short utterance per worker count finds where, the full protocol then runs at the best count; the all-cores
    # figure BASELINE.md asks for is reported from its short sample when it is the slow end of that curve
    short = 8
    sweep = {}
    for w in sorted({4, 8, 16, 32, 64, physical}):
        if w > physical or w <= 2:
            continue
        dt = one(w, short)
        sweep[w] = round(short * FRAME_SEC / dt, 3)
        if dt > 6.0:
            break
    best_w = max(sweep, key=sweep.get) if sweep else 2
    if sweep and sweep[best_w] > legs["reference_default_2_workers"]["xrt"] * short / frames:   # worth the full protocol
        legs["best_worker_count"] = leg(best_w)
    if physical in sweep:
        if physical == best_w and "best_worker_count" in legs:
            legs["all_physical_cores"] = legs["best_worker_count"]
        else:
            legs["all_physical_cores"] = {"workers": physical, "xrt": sweep[physical], "timed_runs": 1, "frames": short,
                                          "note": "short sample: slower than fewer workers (fork-join overhead per operator)"}
    else:
        legs["all_physical_cores"] = {"workers": physical, "xrt": None, "note": f"not run: already {min(sweep.values()) if sweep else 0} x real-time "
                                                                              f"at {max(sweep) if sweep else 0} workers and falling"}
    om.close()
    full = [v for v in legs.values() if v.get("timed_runs") == runs]
    best = max(full, key=lambda v: v["xrt"])
    return {"value": best["xrt"], "unit": "x real-time", "cores": best["workers"], "kind": "port",
            "sample": f"1 warm-up + {runs} timed utterances per leg, one after the other (batch 1: the reference has no batching), 25 tokens on a "
                      f"125-frame voice state, {frames} frames = {frames*FRAME_SEC:.2f} s of audio each, f32 math on the same checkpoint shapes; p50; "
                      f"worker-count sweep on {short}-frame utterances",
            "legs": legs, "worker_sweep_xrt": sweep, "cpu_model": model, "logical_cpus": logical, "usable_cpus": usable, "physical_cores": physical}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="b64_10s_bf16", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-b1", action="store_true")
    ap.add_argument("--no-two-engines", action="store_true", help="skip the two-engine, plain-launch and temperature extras (profiling: concurrent kernels "
                                                                  "stretch each other, and rocprofv3 slows launches that are issued one by one)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 counter passes that fill roofline.traffic")
    ap.add_argument("--cpu-frames", type=int, default=63)
    ap.add_argument("--native-broadcast", action="store_true", help="N > 1: broadcast the weight arena with the library's own RCCL call (ptts_rccl_broadcast)")
    ap.add_argument("--startup-only", action="store_true", help="CPU rehearsal of the N-rank start-up (gloo): plan, fill, ONE broadcast, sharding; no GPU, no timing")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become one.  Nothing here has touched the GPU; the ranks are fresh processes.
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")

    if args.startup_only:
        return startup_only(rank, world)
    global NATIVE_BROADCAST
    NATIVE_BROADCAST = bool(args.native_broadcast)

    traffic = None
    if rank == 0 and world == 1 and not args.no_traffic:
        # before this process initialises the GPU: the counter passes run as child processes of their own
        t0 = time.time()
        try:
            traffic = measure_traffic()
            traffic["_trace"] = trace_pass(WORKLOADS[args.workload]["frames"]) if args.workload == "b64_10s_bf16" else (None, 0, "only for the headline workload")
        except Exception as e:  # noqa: BLE001
            traffic = {"_notes": [f"traffic measurement failed: {e}"]}
        log(f"[bench] PMC traffic passes took {time.time()-t0:.1f}s: { {k: v for k, v in traffic.items() if k == 'k_skinny' or k.startswith('_n')} }")

    import torch  # before the HIP library: both must share one HIP runtime (go-pocket-tts_amd/runtime.py lib())
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py (rank {rank} of {world}): no HIP device visible -- this benchmark needs an MI355X (there is no CPU fallback)")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py (rank {rank} of {world}): LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        barrier = lambda: dist.barrier()
        rccl_ranks = dist.get_world_size()
    else:
        barrier = lambda: None
    sync = torch.cuda.synchronize

    import ptts_amd
    pkg = ptts_amd.load()
    cfg = pkg.synth.SynthConfig.full()
    wl = WORKLOADS[args.workload]
    path = checkpoint_path(pkg, wl["file"], rank, barrier)
    t0 = time.time()
    model, arena = open_model(pkg, path, wl, rank, world, local)
    log(f"[bench] rank {rank}: model resident in {time.time()-t0:.1f}s ({model.info.arena_bytes/1e6:.0f} MB arena, {model.info.n_params/1e6:.1f} M params)")
    voice = model.upload_voice(pkg.VoiceModelState(voice_modules(pkg, cfg)))
    all_prompts = pkg.synth.make_prompts(wl["batch"] * world, 25, 4000, seed=42)
    prompts = shard_prompts(all_prompts, rank, wl["batch"])

    elapsed_own, lat, frames = run_workload(pkg, model, wl, prompts, voice, args.steps, args.warmup, barrier, sync)
    dev = f"cuda:{local}"
    elapsed = max_over_ranks(elapsed_own, world, dev)
    per_rank_audio = wl["batch"] * wl["frames"] * FRAME_SEC * args.steps
    per_rank_xrt = [round(per_rank_audio / e, 1) for e in gather_over_ranks(elapsed_own, world, dev)]
    audio_s = per_rank_audio * world
    result = {
        "metric": "synthesized audio sec/sec (xRT)", "value": round(audio_s / elapsed, 1), "unit": "x real-time",
        "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": wl["dtype"], "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['batch']} utterances/GPU x {wl['frames']} frames "
                               f"({wl['frames']*FRAME_SEC:.1f} s each), 25-token prompt + 125-frame voice state, greedy "
                               f"(temperature 0), hipGraph-captured AR step, Mimi decode to 24 kHz PCM written to host",
                   "batch_per_gpu": wl["batch"], "frames": wl["frames"], "weights": wl["file"], "kv": "bf16" if wl["kv"] else "f32",
                   "checkpoint": "synthetic, shapes of tts_b6369a24 (seed 1234)", "sharding": f"utterances dealt to {world} rank(s); one weight broadcast at init"},
        "p50_utterance_latency_ms": round(1e3 * statistics.median(lat), 2),
        "per_rank_xrt": per_rank_xrt,
    }
    if rank == 0 and world == 1 and not args.no_two_engines:
        # the same workload with the step's kernels launched one by one instead of replayed from the graph (the library's
        # default: no idle gap between replays; reported beside the headline, which keeps BASELINE's named configuration)
        model.set_use_graph(False)
        e2, lat2, _ = run_workload(pkg, model, wl, prompts, voice, args.steps, 1, barrier, sync)
        result["plain_launches"] = {"value": round(wl["batch"] * wl["frames"] * FRAME_SEC * args.steps / e2, 1), "unit": "x real-time",
                                    "ms_per_step": round(1e3 * e2 / args.steps, 3), "p50_utterance_latency_ms": round(1e3 * statistics.median(lat2), 2),
                                    "config": "same workload, use_graph = 0: every kernel of a frame issued by the host thread"}
        model.set_use_graph(True)
        # the reference's default sampling temperature (config.go:99), noise drawn on the device per (request, step)
        e3, lat3, _ = run_workload(pkg, model, wl, prompts, voice, args.steps, 1, barrier, sync, temperature=0.7)
        result["temperature_0p7"] = {"value": round(wl["batch"] * wl["frames"] * FRAME_SEC * args.steps / e3, 1), "unit": "x real-time",
                                     "ms_per_step": round(1e3 * e3 / args.steps, 3),
                                     "config": "same workload at temperature 0.7: N(0,1)*sqrt(T) per (request, step) drawn on the device (flow_lm.go:386-408)"}
    if not args.no_two_engines and args.workload == "b64_10s_bf16":
        # every rank, its own GPU: the serve-mode shape of configs[3] (per-GPU shard of the worker pool); aggregate = sum of the ranks' audio
        # over the slowest rank's wall time, like `value`
        # (a pass that fails on ONE rank must not leave the others waiting in a collective: every rank runs the pass on its own, then all of them agree on
        # whether it succeeded everywhere before any aggregate is formed)
        def everywhere(ok: bool) -> bool:
            return max_over_ranks(0.0 if ok else 1.0, world, dev) == 0.0
        serve = {}
        # (the two-engine batch-at-a-time leg first, as in rounds 2-4: run behind the continuous legs it came out bimodal, 10-11 k or 15.7-16 k x from run to run --
        # which hardware queues its engines' streams land on depends on what was created and destroyed before them: DESIGN.md section 5, the stream / queue rule)
        try:
            sp = serve_pass(pkg, model, wl, prompts, voice, barrier, sync)
        except Exception as e:  # noqa: BLE001
            log(f"[bench] two-engine serve pass failed on rank {rank}: {e}")
            sp = None
        if everywhere(sp is not None):
            a_s, dt_s, p50_s, mb_s = sp
            dt_all = max_over_ranks(dt_s, world, dev)
            serve["two_engines_batch_at_a_time"] = {
                "value": round(a_s * world / dt_all, 1), "unit": "x real-time", "p50_utterance_latency_ms": round(1e3 * p50_s, 1),
                "per_rank_xrt": [round(a_s / e, 1) for e in gather_over_ranks(dt_s, world, dev)], "mean_batch": round(mb_s, 1),
                "config": f"per GPU: one dispatcher (batch-at-a-time collector, window 3 ms) over 2 engines sharing the weight arena, 128 closed-loop clients x 4 "
                          f"requests of {wl['frames']} frames, PCM16 (rounds 2-4's serve_mode figure); {world} GPU(s)"}
        for key, kw, what in (("uniform", dict(slots=256, clients=512, per_client=4, mixed=False), f"512 closed-loop clients x 4 requests of {wl['frames']} frames"),
                              ("mixed_continuous", dict(slots=192, clients=384, per_client=4, mixed=True),
                               "384 closed-loop clients x 4 requests of 25..150 frames (2-12 s, uniformly drawn)")):
            try:
                cp = serve_continuous_pass(pkg, model, wl, voice, barrier, sync, **kw)
            except Exception as e:  # noqa: BLE001
                log(f"[bench] serve pass '{key}' failed on rank {rank}: {e}")
                cp = None
            if everywhere(cp is not None):
                a_m, dt_m, p50_m, occ_m, fb_m = cp
                dt_all = max_over_ranks(dt_m, world, dev)
                a_all = sum(gather_over_ranks(a_m, world, dev))
                serve[key] = {
                    "value": round(a_all / dt_all, 1), "unit": "x real-time", "p50_utterance_latency_ms": round(1e3 * p50_m, 1),
                    "per_rank_xrt": [round(x / dt_all, 1) for x in gather_over_ranks(a_m, world, dev)], "mean_occupied_slots": round(occ_m, 1), "flow_cluster_fallbacks": fb_m,
                    "config": f"per GPU: the dispatcher's default for one model on its GPU -- one continuous-batching engine, {kw['slots']} slots; {what}, PCM16; "
                              f"{world} GPU(s), no exchange between them"}
        if "uniform" in serve:   # serve_mode = uniform traffic through the default configuration; the other legs beside it
            result["serve_mode"] = dict(serve.pop("uniform"), **serve)
        elif serve:
            result["serve_mode"] = serve
    if rank == 0:
        try:
            result["roofline"] = roofline_pass(pkg, model, wl, prompts, voice, traffic)
        except Exception as e:  # noqa: BLE001
            log(f"[bench] roofline pass failed: {e}")
            result["roofline"] = None
    if rank == 0 and world == 1 and not args.no_two_engines and args.workload == "b64_10s_bf16":
        # more rows per AR step through one engine (a step's time hardly depends on its rows): labelled extras beside the unchanged 64-per-GPU headline
        for bw in (128, 256):
            try:
                result[f"b{bw}"] = wide_batch_pass(pkg, model, wl, bw, voice, max(3, args.steps), sync)
            except Exception as e:  # noqa: BLE001
                log(f"[bench] b{bw} pass failed: {e}")
    if rank == 0 and world == 1 and not args.no_b1 and not args.no_two_engines and args.workload == "b64_10s_bf16":
        # serving configuration, reported beside the headline (not `value`): two engines over the same weights, each running
        # the same 64-utterance passes back to back, so that one pass's Mimi decode overlaps the other's prefill + AR loop
        try:
            import threading
            m2 = model.share()
            cfgs2 = gen_cfgs(pkg, wl, len(prompts), voice)
            toks2 = [p.tolist() for p in prompts]
            n2 = max(3, args.steps)

            def worker(m):
                keep = None
                for _ in range(n2 + 1):
                    keep = m.generate_batch(toks2, cfgs2)
                return keep

            for m in (model, m2):
                m.generate_batch(toks2, cfgs2)
            sync()
            t0 = time.perf_counter()
            ts = [threading.Thread(target=worker, args=(m,)) for m in (model, m2)]
            [t.start() for t in ts]
            [t.join() for t in ts]
            sync()
            dt2 = time.perf_counter() - t0
            result["two_engines"] = {"value": round(2 * (n2 + 1) * wl["batch"] * wl["frames"] * FRAME_SEC / dt2, 1), "unit": "x real-time",
                                     "config": f"2 engines over one weight arena (ptts_model_share), {n2 + 1} passes of {wl['batch']} utterances each, concurrently: "
                                               "128 utterances in flight on the GPU"}
            m2.close()
        except Exception as e:  # noqa: BLE001
            log(f"[bench] two-engine pass failed: {e}")
    voice.close()
    model.close()
    del arena

    if rank == 0 and world == 1:
        if not args.no_b1 and args.workload != "b1_5s_f32":
            # BASELINE.json configs[1]: batch 1, f32 weights, 5-s utterance (latency-oriented)
            wl1 = WORKLOADS["b1_5s_f32"]
            p1 = checkpoint_path(pkg, wl1["file"], 0, lambda: None)
            m1, _ = open_model(pkg, p1, wl1, 0, 1, local)
            v1 = m1.upload_voice(pkg.VoiceModelState(voice_modules(pkg, cfg)))
            e1, lat1, _ = run_workload(pkg, m1, wl1, all_prompts[:1], v1, max(3, args.steps), 2, lambda: None, sync)
            n1 = max(3, args.steps)
            result["b1_f32"] = {"value": round(wl1["frames"] * FRAME_SEC * n1 / e1, 1), "unit": "x real-time",
                                "p50_utterance_latency_ms": round(1e3 * statistics.median(lat1), 2),
                                "config": "1 utterance x 63 frames (5.04 s), f32 weights and KV"}
            try:
                result["b1_f32"]["roofline"] = roofline_pass(pkg, m1, wl1, all_prompts[:1], v1)
            except Exception as e:  # noqa: BLE001
                log(f"[bench] b1 roofline pass failed: {e}")
            v1.close()
            m1.close()
        if not args.no_b1 and args.workload == "b64_10s_bf16":
            # N4 (BASELINE.json configs[4] names int8): the headline workload on per-row-scaled int8 step weights (PTTS_WEIGHTS_INT8;
            # half the bytes of the bf16 step), bf16 KV -- what halving the weight bytes buys a launch-latency-bound step
            try:
                wl8 = dict(WORKLOADS["b64_10s_bf16"], weights=pkg.runtime.WEIGHTS_INT8)
                m8, _ = open_model(pkg, path, wl8, 0, 1, local)
                v8 = m8.upload_voice(pkg.VoiceModelState(voice_modules(pkg, cfg)))
                n8 = max(3, args.steps // 2)
                e8, _, _ = run_workload(pkg, m8, wl8, prompts, v8, n8, 1, lambda: None, sync)
                result["int8_weights"] = {"value": round(wl8["batch"] * wl8["frames"] * FRAME_SEC * n8 / e8, 1), "unit": "x real-time",
                                          "ms_per_step": round(1e3 * e8 / n8, 3), "arena_mb": round(m8.info.arena_bytes / 1e6, 1),
                                          "config": "same workload, weights = PTTS_WEIGHTS_INT8 (weight-only, per-row scales, converted to bf16 in registers), bf16 KV"}
                v8.close()
                m8.close()
            except Exception as e:  # noqa: BLE001
                log(f"[bench] int8 pass failed: {e}")
        if not args.no_cpu_baseline:
            try:
                p32 = checkpoint_path(pkg, "F32", 0, lambda: None)
                result["cpu_baseline"] = cpu_baseline(pkg, p32, cfg, args.cpu_frames)
            except Exception as e:  # noqa: BLE001
                log(f"[bench] cpu baseline failed: {e}")
                result["cpu_baseline"] = None
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


def startup_only(rank, world):
    """The N-rank start-up on CPU (gloo): what tests/test_multirank_cpu.py checks through the same launch path the GPU run uses."""
    import torch
    import torch.distributed as dist
    import ptts_amd
    pkg = ptts_amd.load()
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    path = os.path.join(tempfile.gettempdir(), f"ptts_bench_{os.getuid()}_startup_tiny.safetensors")
    if rank == 0 and not os.path.exists(path):
        tmp = path + f".tmp{os.getpid()}"
        pkg.synth.write_safetensors(tmp, pkg.synth.make_checkpoint(pkg.synth.SynthConfig.tiny(), seed=1234))
        os.replace(tmp, path)
    if world > 1:
        dist.barrier()
    plan, nbytes = pkg.Model.plan(path, weights=pkg.WEIGHTS_BF16)
    arena = torch.zeros(nbytes, dtype=torch.uint8)
    if rank == 0:
        arena.copy_(torch.from_numpy(pkg.Model.plan_fill_host(plan, nbytes)))
    if world > 1:
        dist.broadcast(arena, src=0)
    same = bool(np.array_equal(arena.numpy(), pkg.Model.plan_fill_host(plan, nbytes)))
    pkg.Model.plan_free(plan)
    allp = pkg.synth.make_prompts(4 * world, 25, 64, seed=42)
    mine = shard_prompts(allp, rank, 4)
    t = max_over_ranks(1.0 + rank, world)
    per = gather_over_ranks(1.0 + rank, world)
    ranks = dist.get_world_size() if world > 1 else 1
    ok = same and t == float(world) and per == [1.0 + r for r in range(world)] and len(mine) == 4
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"startup_only": True, "n_gpus": world, "rccl_ranks": ranks, "arena_bytes": nbytes, "arena_matches_local_fill": same, "ok": ok}), flush=True)
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
