#!/usr/bin/env python3
"""Headline benchmark: synthesized audio seconds per wall second (xRT) of the MI355X PocketTTS path.

One "step" = one pass of the hot path over one batch of synthetic utterance chunks:
tts.Runtime.GenerateAudio for 64 concurrent utterances per GPU (text prompt of 25 tokens on top of a
125-frame voice model state, prefill, 125 autoregressive steps = 10 s of audio each, Mimi decode to
24 kHz PCM).  BASELINE.json configs[2] at N=1 and configs[3] (64 per GPU) at N=8: weak scaling, no
data-path collective -- one RCCL broadcast of the weight arena at start-up.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints one JSON line.  `roofline` prices the dominant kernel (the AR step's weight-streaming linear)
against HBM; `cpu_baseline` times the CPU oracle (a C restatement of the reference's AVX2/FMA path: the Go
binary itself cannot be built here) on the host cores, on a bounded sample, on rank 0 at N=1 only.
"""
import argparse
import json
import os
import statistics
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch  # before the HIP library: both must share one HIP runtime (go-pocket-tts_amd/runtime.py lib())

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FRAME_SEC = 0.08           # 1920 samples @ 24 kHz (PLAN.md:37)

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
        tmp = path + ".tmp"
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
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def open_model(pkg, path, wl, rank, world, device):
    """Two-phase open: rank 0 fills the device arena, one RCCL broadcast hands it to the other ranks."""
    kw = dict(device=device, weights=wl["weights"], kv=wl["kv"], max_batch=wl["batch"], use_graph=True)   # configs[2]: hipGraph-captured step
    if world == 1:
        return pkg.Model.open(path, **kw), None
    import torch.distributed as dist
    plan, nbytes = pkg.Model.plan(path, **kw)
    arena = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{device}")
    if rank == 0:
        model = pkg.Model.open_planned(plan, arena.data_ptr(), fill=True)
        torch.cuda.synchronize()
    dist.broadcast(arena, src=0)          # the only collective of the whole job (weights, once)
    torch.cuda.synchronize()
    if rank != 0:
        model = pkg.Model.open_planned(plan, arena.data_ptr(), fill=False)
    return model, arena


def run_workload(pkg, model, wl, prompts, voice, steps, warmup, barrier, sync):
    cfgs = [pkg.RuntimeGenerateConfig(temperature=0.0, eos_threshold=float("inf"), max_steps=wl["frames"],
                                      lsd_decode_steps=1, frames_after_eos=3, device_voice=voice) for _ in range(len(prompts))]
    toks = [np.ascontiguousarray(p, np.int64) for p in prompts]
    out = None
    for _ in range(warmup):
        out = model.generate_batch(toks, cfgs)   # held like in the timed loop: the pinned result pool reaches its steady state (two sets)
    sync(); barrier()
    lat = []
    t0 = time.perf_counter()
    for _ in range(steps):
        s0 = time.perf_counter()
        out = model.generate_batch(toks, cfgs)
        lat.append(time.perf_counter() - s0)
    sync(); barrier()
    elapsed = time.perf_counter() - t0
    log("[bench] per-step ms: " + " ".join(f"{1e3*x:.1f}" for x in lat))
    frames = sum(o.n_frames for o in out)
    assert all(o.n_frames == wl["frames"] and o.pcm.shape[0] == wl["frames"] * 1920 for o in out)
    assert all(np.isfinite(o.pcm).all() for o in out[:2])
    return elapsed, lat, frames


def roofline_pass(pkg, model, wl, prompts, voice):
    """One extra pass with HIP events around every launch of the dominant kernel (eager, same stream)."""
    cfgs = [pkg.RuntimeGenerateConfig(temperature=0.0, eos_threshold=float("inf"), max_steps=wl["frames"],
                                      lsd_decode_steps=1, device_voice=voice) for _ in range(len(prompts))]
    toks = [p.tolist() for p in prompts]
    model.profile_enable(True)
    try:
        model.generate_batch(toks, cfgs)
        prof = model.profile_read()
    finally:
        model.profile_enable(False)
    achieved = prof["algorithmic_bytes"] / (prof["total_ms"] * 1e-3) / 1e9 if prof["total_ms"] > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # filled from separate rocprofv3 --pmc passes
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(prof["kernel"], {}).get(wl_name(wl))
        except Exception:  # noqa: BLE001
            traffic = None
    return {"bound": "hbm", "kernel": prof["kernel"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "launches": prof["launches"],
            "avg_launch_us": round(prof["total_ms"] * 1e3 / max(1, prof["launches"]), 3),
            "algorithmic_bytes_per_launch": round(prof["algorithmic_bytes"] / max(1, prof["launches"]))}


def wl_name(wl):
    return next(k for k, v in WORKLOADS.items() if v is wl)


def cpu_baseline(pkg, path, cfg, budget_frames=16):
    """The CPU oracle (C restatement of the reference's AVX2/FMA path) on the host cores, batch 1, bounded sample."""
    from oracle import oracle as O
    O.build()
    om = O.OracleModel.from_file(path)
    mods = voice_modules(pkg, cfg)
    toks = pkg.synth.make_prompts(1, 25, 4000, seed=42)[0]
    workers = 2   # reference defaults: conv-workers 2, runtime-workers falls back to it (config.go:76-83, service.go:318-328)
    O.set_workers(workers, workers)
    O.set_use_avx2(True)
    # bounded sample: whole utterances of `budget_frames` frames, one after the other, until >= 12 s of CPU work (<= 30 s)
    frames, n_utt, t0 = 0, 0, time.perf_counter()
    while True:
        r = om.generate(toks, max_steps=budget_frames, eos_threshold=1e30, frames_after_eos=3, voice_state=mods)
        frames += r["n_frames"]
        n_utt += 1
        dt = time.perf_counter() - t0
        if dt >= 12.0 or dt * (n_utt + 1) / n_utt > 30.0:
            break
    om.close()
    return {"value": round(frames * FRAME_SEC / dt, 3), "unit": "x real-time", "cores": workers, "kind": "port",
            "sample": f"{n_utt} utterances one after the other (batch 1, the reference has no batching), 25 tokens on a 125-frame voice state, "
                      f"{budget_frames} frames = {budget_frames*FRAME_SEC:.2f} s of audio each, f32 math on the same checkpoint, "
                      f"{dt:.1f} s wall; host has {os.cpu_count()} logical CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="b64_10s_bf16", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-b1", action="store_true")
    ap.add_argument("--no-two-engines", action="store_true", help="skip the two-engine and plain-launch extras (profiling: concurrent kernels stretch each other, "
                                                                  "and rocprofv3 slows launches that are issued one by one)")
    ap.add_argument("--cpu-frames", type=int, default=63)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus={args.gpus}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        barrier = lambda: dist.barrier()
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

    elapsed, lat, frames = run_workload(pkg, model, wl, prompts, voice, args.steps, args.warmup, barrier, sync)
    elapsed = max_over_ranks(elapsed, world, f"cuda:{local}")
    audio_s = wl["batch"] * world * wl["frames"] * FRAME_SEC * args.steps
    result = {
        "metric": "synthesized audio sec/sec (xRT)", "value": round(audio_s / elapsed, 1), "unit": "x real-time",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": wl["dtype"], "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['batch']} utterances/GPU x {wl['frames']} frames "
                               f"({wl['frames']*FRAME_SEC:.1f} s each), 25-token prompt + 125-frame voice state, greedy "
                               f"(temperature 0), hipGraph-captured AR step (46 kernels per frame, five frames per replay; one graph per attention round count), Mimi decode to 24 kHz PCM written to host",
                   "batch_per_gpu": wl["batch"], "frames": wl["frames"], "weights": wl["file"], "kv": "bf16" if wl["kv"] else "f32",
                   "checkpoint": "synthetic, shapes of tts_b6369a24 (seed 1234)", "sharding": f"utterances dealt to {world} rank(s); one weight broadcast at init"},
        "p50_utterance_latency_ms": round(1e3 * statistics.median(lat), 2),
    }
    if rank == 0 and world == 1 and not args.no_two_engines:
        # the same workload with the step's kernels launched one by one instead of replayed from the graph (the library's
        # default: no idle gap between replays; reported beside the headline, which keeps BASELINE's named configuration)
        model.set_use_graph(False)
        e2, lat2, _ = run_workload(pkg, model, wl, prompts, voice, args.steps, 1, barrier, sync)
        result["plain_launches"] = {"value": round(wl["batch"] * wl["frames"] * FRAME_SEC * args.steps / e2, 1), "unit": "x real-time",
                                    "ms_per_step": round(1e3 * e2 / args.steps, 3), "p50_utterance_latency_ms": round(1e3 * statistics.median(lat2), 2),
                                    "config": "same workload, use_graph = 0: 46 launches per frame issued by the host thread"}
    if rank == 0:
        try:
            result["roofline"] = roofline_pass(pkg, model, wl, prompts, voice)
        except Exception as e:  # noqa: BLE001
            log(f"[bench] roofline pass failed: {e}")
            result["roofline"] = None
    if rank == 0 and world == 1 and not args.no_b1 and not args.no_two_engines and args.workload == "b64_10s_bf16":
        # serving configuration, reported beside the headline (not `value`): two engines over the same weights, each running
        # the same 64-utterance passes back to back, so that one pass's Mimi decode overlaps the other's prefill + AR loop
        try:
            import threading
            m2 = model.share()
            cfgs2 = [pkg.RuntimeGenerateConfig(temperature=0.0, eos_threshold=float("inf"), max_steps=wl["frames"], lsd_decode_steps=1,
                                               frames_after_eos=3, device_voice=voice) for _ in prompts]
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
        if not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(pkg, path, cfg, args.cpu_frames)
            except Exception as e:  # noqa: BLE001
                log(f"[bench] cpu baseline failed: {e}")
                result["cpu_baseline"] = None
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
