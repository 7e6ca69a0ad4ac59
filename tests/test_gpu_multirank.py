"""N > 1 ranks on real GPUs -- runs only where the box has at least two (the round's one-GPU boxes skip it; the driver's 8-GPU node does not).
SURVEY.md 8(e): one process per GPU, ONE weight broadcast at init over RCCL / xGMI, no per-step collective; the worker-pool shape it
mirrors is internal/server/server.go:119-143,398-421.  bench.py checks by itself that every rank's arena holds the same bytes after the
broadcast (a 64-bit sum compared across ranks) and that the launcher's rank count is the one asked for."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus() -> int:
    import torch
    return torch.cuda.device_count()   # (does not initialise the GPU on this image)


@pytest.mark.parametrize("native", [False, True])
def test_two_ranks_share_one_broadcast_arena(native):
    """native=True: the library's own ncclBroadcast (ptts_rccl_unique_id / ptts_rccl_broadcast, csrc/broadcast.cpp) -- the path a host
    without PyTorch (the reference's Go server) takes; until a box with two GPUs runs this it has only ever run with ONE rank.
    A plain test (skipped below two GPUs): its first real run can fail."""
    if _gpus() < 2:
        pytest.skip("needs two GPUs")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-b1", "--no-two-engines", "--no-traffic"]
    if native:
        cmd.append("--native-broadcast")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import signal
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    try:
        out, err = p.communicate(timeout=420)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)   # exactly the process group started above (bench.py and the ranks it spawned)
        p.communicate()
        pytest.fail("bench.py --gpus 2 did not finish in 7 minutes")
    assert p.returncode == 0, (out[-2000:], err[-4000:])
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == line["rccl_ranks"] == 2 and len(line["per_rank_xrt"]) == 2 and line["scaling"] == "weak"
    assert line["value"] > 1.5 * min(line["per_rank_xrt"])   # two ranks' audio over the slower rank's time


def _one_process_n_models(pkg, devices):
    """One process, one model per entry of `devices` (the first opens the file, the others are ptts_model_replicate copies over the GPUs' direct link), ONE
    dispatcher over all of them, 4 x len(devices) x 8 clients: every caller's audio equals the stand-alone audio of the first model (rows of a batch never
    mix, every GPU holds the same weights), every model served batches, and the replicas survive the closing of the model they were copied from."""
    import threading
    import numpy as np
    cfg = pkg.synth.SynthConfig.tiny()
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "tiny.safetensors")
        pkg.synth.write_safetensors(path, pkg.synth.make_checkpoint(cfg, seed=99))
        base = pkg.Model.open(path, device=devices[0], max_batch=8)
        models = [base] + [base.replicate(d) for d in devices[1:]]
        assert [m.info.arena_bytes for m in models] == [base.info.arena_bytes] * len(models)
        prompts = [p.tolist() for p in pkg.synth.make_prompts(16, 7, cfg.n_bins, seed=3)]
        c = pkg.RuntimeGenerateConfig(max_steps=5, eos_threshold=float("inf"), frames_after_eos=3, want_latents=True)
        want = base.generate_batch(prompts[:8], [c] * 8) + base.generate_batch(prompts[8:], [c] * 8)
        for m in models[1:]:                               # a replica alone gives the bits of the original: the same kernels on the same bytes
            got = m.generate_batch(prompts[:8], [c] * 8)
            for a, b in zip(got, want[:8]):
                assert np.array_equal(a.latents, b.latents) and np.array_equal(a.pcm, b.pcm)
        d = pkg.Dispatcher(models, max_batch=8, window_us=2000, continuous=False)   # (the batch collector: on one GPU per model the default would be continuous batching)
        n_clients = 4 * len(models) * 8
        res, errs = [None] * n_clients, [None] * n_clients

        def client(i):
            try:
                res[i] = d.generate(prompts[i % 16], c)
            except Exception as e:  # noqa: BLE001
                errs[i] = e

        ts = [threading.Thread(target=client, args=(i,)) for i in range(n_clients)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        st = d.stats()
        d.close()
        assert not any(errs), [str(e) for e in errs if e][:3]
        for i in range(n_clients):
            assert res[i].n_frames == want[i % 16].n_frames
            # (a dispatcher batch is whatever arrived together: its packed prefill may take other GEMM tilings than the reference batches of eight did --
            # kernel-selection rounding over five free-running steps, the tolerance of the other dispatcher tests: tests/test_gpu_model.py)
            np.testing.assert_allclose(res[i].latents, want[i % 16].latents, rtol=0, atol=1e-4 * max(1.0, float(np.abs(want[i % 16].latents).max())))
        assert st["requests"] == n_clients and st["batches"] >= len(models), st
        base.close()                                       # the replicas own their arenas
        for m in models[1:]:
            got = m.generate_batch(prompts[:8], [c] * 8)
            for a, b in zip(got, want[:8]):
                assert np.array_equal(a.latents, b.latents)
            m.close()


def test_one_process_dispatcher_over_replicas_on_one_gpu(pkg):
    """The one-process / N-model start-up (ptts_model_replicate + one dispatcher: the shape of the reference's single server process,
    internal/server/server.go:119-143,398-421, cmd/pockettts/serve.go:15-52) with both models on device 0: everything but the cross-GPU link of
    hipMemcpyPeer, on a one-GPU box."""
    _one_process_n_models(pkg, [0, 0])


def test_one_process_dispatcher_over_models_on_every_gpu(pkg):
    """The same over every GPU of the box (device 0's arena handed to devices 1..N-1 by hipMemcpyPeer, no RCCL): needs two GPUs."""
    n = _gpus()
    if n < 2:
        pytest.skip("needs two GPUs")
    _one_process_n_models(pkg, list(range(min(n, 8))))
