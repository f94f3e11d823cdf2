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


# ---- the stream driver: contiguous slices + one-frame halo at the seams (uwimageproc_amd/stream.py) -----------------
def _fake_process_factory():
    """A CPU stand-in for FramePipe with the same chaining contract: out = 255 - frame, ratio_i = mean|out_i - out_{i-1}|
    (first frame of the stream against itself), params from the frame's first pixel."""
    import numpy as np
    state = {"prev": None}

    def proc(batch, first):
        if first:
            state["prev"] = None
        out = 255 - batch
        ratio, par = [], []
        for j in range(batch.shape[0]):
            prev = out[j] if state["prev"] is None else state["prev"]
            ratio.append(float(np.abs(out[j].astype(np.int32) - prev.astype(np.int32)).mean()))
            par.append((int(batch[j, 0, 0, 0]) % 5, int(batch[j, 0, 0, 1]) % 25))
            state["prev"] = out[j]
        return out, ratio, par
    return proc


def _frame(i):
    import numpy as np
    rng = np.random.default_rng(100 + i)
    return rng.integers(0, 256, (6, 8, 3), dtype=np.uint8)


def _stream_worker(rank, world, port, n_frames, batch, q):
    import torch.distributed as dist
    from uwimageproc_amd import stream
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    drv = stream.StreamDriver(n_frames, rank, world, batch, _fake_process_factory())
    seen = []
    ratios, params = drv.run(_frame, sink=lambda i, f: seen.append(i))
    allr, allp = drv.gather(ratios, params)
    q.put((rank, seen, allr, allp, drv.frame_indices()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,batch", [(7, 3), (8, 4), (5, 8)])
def test_stream_driver_two_ranks_equals_one(n_frames, batch):
    """Two gloo ranks, each on its slice with a one-frame halo, reproduce the single-rank stream value for value."""
    from uwimageproc_amd import stream
    one = stream.StreamDriver(n_frames, 0, 1, batch, _fake_process_factory())
    r1, p1 = one.run(_frame)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, n_frames, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res.sort()
    a, b = sharding.frame_slice(n_frames, 1, 2)
    assert res[0][1] == list(range(0, a)) and res[1][1] == list(range(a, b))       # each frame delivered once, by its owner
    assert res[1][4][0] == a - 1                                                     # rank 1 read its predecessor frame
    for _, _, allr, allp, _ in res:
        assert allr == r1 and allp == p1
    assert stream.select_from_ratios([0.9, 0.3, -2.0, 0.5], 0.4) == [1]
    assert stream.select_from_ratios([0.9, 0.3, -2.0, 0.5], 0.41) == [1, 2]
