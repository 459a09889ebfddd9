"""N > 1 on the device: bench.py launched the way the driver launches it, with two ranks.

* two visible GPUs: `backend="nccl"` (RCCL), one rank per GPU, the pose records gathered device to device -- skipped on a one-GPU
  box, which is all this pool gives a round's own runs;
* one GPU: the rehearsal mode (`FLOPE_BENCH_REHEARSE=1`: both ranks on cuda:0, gloo, poses gathered through host memory) -- the
  same control flow (shard, barrier, max-over-ranks timing, gather, one JSON line from rank 0), never a reported number.
No scaling curve has been measured on hardware: SCALE runs are the driver's."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench_two_ranks(extra_env, outer_launcher=True):
    """outer_launcher: the driver's documented form (`python -m torch.distributed.run ... bench.py --gpus 2`); without it bench.py
    is started bare (`python bench.py --gpus 2`, no WORLD_SIZE) and must start its own ranks as a child process."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())] if outer_launcher else []
    cmd = [sys.executable] + launcher + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                                         "--no-cpu-baseline", "--no-alt"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # rank 0 prints ONE line
    return json.loads(lines[0])


def _check(out):
    assert out["metric"] == "poses_per_sec" and out["unit"] == "poses/s" and out["n_gpus"] == 2 and out["steps"] == 3
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 512 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and abs(out["value"] - 2 * 3 * 256 / (out["ms_per_step"] * 3e-3)) <= 1e-3 * out["value"]
    assert out["rot_err_vs_oracle"]["max_abs_R"] <= 1e-3


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL between two ranks)")
def test_bench_two_ranks_over_rccl():
    _check(_bench_two_ranks({}))


@pytest.mark.parametrize("outer_launcher", [True, False])
def test_bench_two_ranks_rehearsed_on_one_gpu(outer_launcher):
    out = _bench_two_ranks({"FLOPE_BENCH_REHEARSE": "1"}, outer_launcher)
    _check(out)
    assert "REHEARSAL" in out["data"]
