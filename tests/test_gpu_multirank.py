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
