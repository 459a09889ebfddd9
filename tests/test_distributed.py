"""CPU, world_size 2 over gloo: the N>1 plumbing of the data-parallel driver (shard, gather,
max-over-ranks timing) -- the same code bench.py runs over RCCL."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from flope_amd import distributed as D
    r, w, _ = D.init_from_env("gloo")
    lo, hi = D.shard_range(n_items, r, w)
    # pose record i is filled with i so the gathered order can be checked exactly
    local = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, D.POSE_FLOATS)
    counts = [D.shard_range(n_items, k, w)[1] - D.shard_range(n_items, k, w)[0] for k in range(w)]
    allp = D.gather_poses(local, counts)
    D.barrier()
    t = D.max_over_ranks(float(rank + 1), "cpu")
    q.put((rank, allp[:, 0].tolist(), tuple(allp.shape), t))
    torch.distributed.destroy_process_group()


def _run(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_shard_range_partitions():
    from flope_amd.distributed import shard_range
    for n in (0, 1, 7, 64, 65536):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(65536, 3, 8) == (3 * 8192, 4 * 8192)          # BASELINE cfg4: 8 x 8192


def test_gather_equal_shards_world2():
    for rank, col, shape, t in _run(64):
        assert shape == (64, 16) and col == [float(i) for i in range(64)] and t == 2.0


def test_gather_ragged_shards_world2():
    for rank, col, shape, t in _run(7):
        assert shape == (7, 16) and col == [float(i) for i in range(7)]
