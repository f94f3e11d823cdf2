"""CPU: the N > 1 path (frame partition + ordered gather + max-over-ranks) with two gloo ranks."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uwimageproc_amd import sharding  # noqa: E402


def test_slices_cover_and_balance():
    for n in (0, 1, 7, 32, 33, 1000):
        for w in (1, 2, 3, 8):
            sl = sharding.all_slices(n, w)
            assert sl[0][0] == 0 and sl[-1][1] == n
            assert all(sl[i][1] == sl[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in sl]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.frame_slice(4, 2, 2)


def _worker(rank, world, port, n_frames, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = sharding.frame_slice(n_frames, rank, world)
    local = [float(f) * 0.5 + 1.0 for f in range(a, b)]           # a per-frame "overlap ratio"
    full = sharding.gather_in_frame_order(local, n_frames, rank, world)
    slow = sharding.max_over_ranks(1.0 + rank)
    q.put((rank, full, slow))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_and_max():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_frames, world = 7, 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = [float(f) * 0.5 + 1.0 for f in range(n_frames)]
    for rank, full, slow in res:
        assert full == expect           # every rank sees all frames, in frame order
        assert slow == 2.0              # max over ranks
