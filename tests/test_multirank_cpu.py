"""N > 1 path on CPU (gloo, world_size 2): the start-up protocol bench.py uses on GPUs -- every rank plans the arena from
the file header, rank 0 fills it, ONE broadcast hands it over -- plus the utterance sharding and the max-over-ranks
timing reduction.  No GPU, no compute through the HIP library (only header planning and the host arena image)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, path, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ptts_amd
    import bench
    pkg = ptts_amd.load()
    # 1. every rank plans from the header alone and must agree on the layout size
    plan, nbytes = pkg.Model.plan(path, weights=pkg.WEIGHTS_BF16)
    sizes = [None] * world
    dist.all_gather_object(sizes, nbytes)
    assert len(set(sizes)) == 1 and nbytes > 0
    # 2. rank 0 fills, one broadcast, the others adopt: identical bytes to a local fill
    arena = torch.zeros(nbytes, dtype=torch.uint8)
    if rank == 0:
        arena.copy_(torch.from_numpy(pkg.Model.plan_fill_host(plan, nbytes)))
    dist.broadcast(arena, src=0)
    local = pkg.Model.plan_fill_host(plan, nbytes)
    assert np.array_equal(arena.numpy(), local)
    pkg.Model.plan_free(plan)
    # 3. sharding: disjoint, complete, no exchange
    allp = pkg.synth.make_prompts(4 * world, 25, 64, seed=42)
    mine = bench.shard_prompts(allp, rank, 4)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine.tolist())
    assert np.array_equal(np.concatenate([np.array(g) for g in gathered]), allp)
    # 4. whole-job time = max over ranks
    t = bench.max_over_ranks(1.0 + rank, world)
    assert t == float(world)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


def test_two_rank_startup_protocol(tmp_path):
    sys.path.insert(0, ROOT)
    import ptts_amd
    pkg = ptts_amd.load()
    path = str(tmp_path / "tiny.safetensors")
    pkg.synth.write_safetensors(path, pkg.synth.make_checkpoint(pkg.synth.SynthConfig.tiny(), seed=1234))
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), path, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def _run_bench(*args):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, env=env)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (fresh children; the parent never touches a
    GPU): the CPU rehearsal runs the start-up protocol through exactly that path and reports n_gpus == rccl_ranks == 2."""
    import json
    r = _run_bench("--gpus", "2", "--startup-only")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == line["rccl_ranks"] == 2 and line["ok"] and line["arena_matches_local_fill"]


def test_bench_without_gpus_fails_fast_from_every_rank():
    """On a box without GPUs the two children say so and the parent exits non-zero: no silent one-GPU run labelled --gpus 2."""
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs")
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "rank 0 of 2" in r.stderr and "rank 1 of 2" in r.stderr and "no HIP device" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_a_rank_count_that_disagrees_with_the_launcher():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


def test_a_serve_pass_that_fails_on_one_rank_still_reaches_both_barriers():
    """bench.py's serve legs run on every rank between two barriers; a rank whose engine fails (prepare or the timed round) must reach both anyway, or the
    other ranks wait in the collective for ever (bench.two_barrier_pass)."""
    sys.path.insert(0, ROOT)
    import bench
    for fail_in in (None, "prepare", "timed"):
        calls = []

        def prepare():
            calls.append("prepare")
            if fail_in == "prepare":
                raise RuntimeError("engine did not come up")

        def timed():
            calls.append("timed")
            if fail_in == "timed":
                raise RuntimeError("a client failed")

        dt, err = bench.two_barrier_pass(lambda: calls.append("barrier"), lambda: calls.append("sync"), prepare, timed)
        assert calls.count("barrier") == 2, (fail_in, calls)
        assert (err is None) == (fail_in is None)
        assert ("timed" in calls) == (fail_in != "prepare")        # nothing is timed on a rank whose preparation failed
        assert calls[-1] == "barrier"


def test_client_threads_are_all_joined_and_a_failure_in_one_is_reported():
    """bench.run_client_threads: the closed-loop clients of the serve legs; a client that fails (or a thread that cannot be started) fails the leg on that rank
    only after every started thread has ended -- nothing keeps calling into a dispatcher that is about to be closed."""
    sys.path.insert(0, ROOT)
    import bench
    import threading
    import time
    seen, lock = [], threading.Lock()

    def client(i, n):
        time.sleep(0.01 * (i % 3))
        with lock:
            seen.append((i, n))
        if i == 5:
            raise ValueError("client 5 failed")

    with pytest.raises(ValueError, match="client 5"):
        bench.run_client_threads(client, 12, 2)
    assert sorted(seen) == [(i, 2) for i in range(12)]          # everybody ran to the end before the failure was reported
    seen.clear()
    bench.run_client_threads(lambda i, n: seen.append(i), 4, 1)
    assert sorted(seen) == [0, 1, 2, 3]
