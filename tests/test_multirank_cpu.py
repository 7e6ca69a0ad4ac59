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
