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


def test_rank_affinity_from_sysfs(tmp_path, monkeypatch):
    """uwimageproc_amd.affinity: the CPUs next to a rank's GPU come from the KFD topology + PCI sysfs (no GPU call);
    two ranks on one NUMA node split its CPUs; a hidden sysfs degrades to "unavailable"."""
    import os
    from uwimageproc_amd import affinity
    assert affinity.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    kfd, pci = tmp_path / "kfd", tmp_path / "pci"
    have = sorted(os.sched_getaffinity(0))
    half = max(1, len(have) // 2)
    lists = [have[:half], have[:half], have[half:] or have[:1]]          # GPUs 0 and 1 share node 0
    # node 0 is a CPU node; nodes 1..3 are GPUs at 0000:0a:00.0, 0000:1b:00.0, 0001:c3:00.0
    locs = [(0, 0, 0), (0x0a00, 0, 1), (0x1b00, 0, 1), (0xc300, 1, 1)]
    for i, (loc, dom, gpu) in enumerate(locs):
        d = kfd / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if gpu else 64}\nsimd_count {1024 if gpu else 0}\nlocation_id {loc}\ndomain {dom}\n")
    for j, bdf in enumerate(("0000:0a:00.0", "0000:1b:00.0", "0001:c3:00.0")):
        d = pci / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{0 if j < 2 else 1}\n")
        (d / "local_cpulist").write_text(",".join(str(c) for c in lists[j]) + "\n")
    kw = dict(kfd_nodes=str(kfd), pci_devices=str(pci), env={})
    assert affinity.gpu_pci_addresses(str(kfd)) == ["0000:0a:00.0", "0000:1b:00.0", "0001:c3:00.0"]
    assert affinity.gpu_local_cpus(2, **kw)["numa_node"] == 1
    assert affinity.gpu_local_cpus(1, kfd_nodes=str(kfd), pci_devices=str(pci), env={"HIP_VISIBLE_DEVICES": "2,0"})["pci"] == "0000:0a:00.0"
    assert "error" in affinity.gpu_local_cpus(5, **kw)
    assert "error" in affinity.gpu_local_cpus(0, kfd_nodes=str(tmp_path / "nope"), pci_devices=str(pci), env={})
    before = os.sched_getaffinity(0)
    try:
        a = affinity.pin_rank_to_gpu(0, ranks_on_node=3, local_rank=0, **kw)
        got0 = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, before)
        b = affinity.pin_rank_to_gpu(1, ranks_on_node=3, local_rank=1, **kw)
        got1 = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, before)
        assert a["pinned"] and b["pinned"]
        assert set(got0) <= set(lists[0]) and set(got1) <= set(lists[1])
        if half >= 2:
            assert not (set(got0) & set(got1))                               # the shared node's CPUs are split
        c = affinity.pin_rank_to_gpu(0, kfd_nodes=str(tmp_path / "nope"), pci_devices=str(pci), env={})
        assert not c["pinned"] and sorted(os.sched_getaffinity(0)) == sorted(before)
    finally:
        os.sched_setaffinity(0, before)


# ---- the key-frame selector across ranks (SURVEY 8e option 1) ----------------------------------------------------------
class _OracleBackend:
    """CPU stand-in for selector.GpuBackend: the oracle's detect / describe / blur per frame, its match + homography +
    overlapArea per (key, frame) pair."""

    def __init__(self, orc, vw, vh, seed=1):
        self.orc, self.vw, self.vh, self.seed = orc, vw, vh, seed

    def extract(self, frames):
        out = []
        for f in frames:
            kps, desc, _ = self.orc.detect_describe(self.orc.resize_gray(f))
            out.append((kps, desc, self.orc.calcBlur(self.orc.resize_bgr(f))))
        return out

    def overlaps(self, key, objs):
        orc, res = self.orc, []
        for kq, dq, _ in objs:
            kt, dt, _ = key
            r = -2.0
            if len(kq) and len(kt):
                idx, dist = orc.match_knn2(dq, dt)
                gq, gt = orc.ratio_test(idx, dist, len(dt))
                if len(gq) >= 4:
                    n, H = orc.find_homography(kq["x"][gq], kq["y"][gq], kt["x"][gt], kt["y"][gt], 640, self.wh[0], seed=self.seed)
                    if n > 0:
                        r = orc.overlapArea(H, self.vw, self.vh)[0]
            res.append(r)
        return res


def _reference_loop(orc, frames, p, k, vw, vh):
    """main.cpp:300-394 frame by frame on the oracle's calcOverlap / calcBlur (what tests/test_cli.py expects of the CLI)"""
    n = len(frames)
    exp = [(0, 0)]
    key, nxt, read = frames[0], 1, 1
    while nxt < n:
        f = frames[nxt]; nxt += 1; read += 1
        ov, _, _ = orc.calcOverlap(key, f, vw, vh, seed=1)
        if ov == -2.0:
            ov = 0.41
        if ov <= p:
            best, bestn, bf = orc.calcBlur(orc.resize_bgr(f)), nxt - 1, f
            eof = False
            for _ in range(k):
                if nxt >= n:
                    eof = True
                    break
                g = frames[nxt]; nxt += 1; read += 1
                b = orc.calcBlur(orc.resize_bgr(g))
                if b > best:
                    best, bestn, bf = b, read, g
            key = bf
            exp.append((len(exp), bestn))
            if eof:
                break
    return exp


def _selector_rank(rank, world, port, q):
    import os
    import sys
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)})
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    import _oracle
    from uwimageproc_amd import selector, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = _oracle.load()
    frames = synth.uw_stream(0, 14, 480, 640, step_frac=0.05)
    be = _OracleBackend(orc, 640, 480)
    be.wh = orc.resize_dims(480, 640)
    rows = selector.select_distributed(be, lambda i: frames[i], len(frames), rank, world, minOverlap=0.7, kWindow=2, batch=3, lookahead=4)
    q.put((rank, [(a, b) for a, b, _, _ in rows]))
    dist.destroy_process_group()


def test_keyframe_selector_across_two_ranks_equals_the_sequential_loop():
    """Two gloo ranks extract their slices, rank 0 replays the decision chain with speculative look-ahead: the exported
    (ID, Frame) rows equal the frame-by-frame loop of main.cpp:300-394 -- the same expectation tests/test_cli.py holds
    the single-GPU CLI to."""
    import multiprocessing as mp
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _oracle
    from uwimageproc_amd import selector, synth
    orc = _oracle.load()
    frames = synth.uw_stream(0, 14, 480, 640, step_frac=0.05)
    exp = _reference_loop(orc, frames, 0.7, 2, 640, 480)
    assert len(exp) >= 3
    # single process, several look-ahead depths: speculation never changes the rows
    be = _OracleBackend(orc, 640, 480)
    be.wh = orc.resize_dims(480, 640)
    recs = be.extract(frames)
    for la in (1, 3, 8, 32):
        assert [(a, b) for a, b, _, _ in selector.chain(recs, be.overlaps, 0.7, 2, la)] == exp, la
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611
    procs = [ctx.Process(target=_selector_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == exp and got[1] == exp
