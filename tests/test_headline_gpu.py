"""GPU: what bench.py TIMES is also CHECKED (task 1).

  (a) the bench's own arrangement -- bench.Rig: 8 sub-batch pipes on 8 streams, 1920x1080 frames of the bench's own
      motion stream (bench.synth_frames), two steps -- every sub-batch against the oracle chain (re-synchronised after
      the dehaze stage, as test_pipeline_gpu.py does: the float64 dehaze agrees to 1e-9 and its 8-bit cast only at exact
      rounding ties) and against each other;
  (b) k_ov_match on BASELINE config 4's size -- 64 pairs x (2048 x 2048) random 486-bit descriptors -- and on ragged
      counts, exact (index, distance) against the oracle's kNN(2).
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

import _knee_mirror as knee_mirror

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from uwimageproc_amd import videostrip as vs  # noqa: E402
from uwimageproc_amd.pipeline import FramePipe  # noqa: E402

pytestmark = pytest.mark.gpu


def test_bench_arrangement_8_streams_1080p_vs_oracle(orc):
    import bench
    S, Fs, H, W = 8, 2, 1080, 1920
    dev = torch.device("cuda", 0)
    rig = bench.Rig(0, dev, S * Fs, H, W, S, 1234)
    rig.run_steps(2)
    rig.drain()
    outs = [p.work.cpu().numpy().copy() for p in rig.pipes]
    ratios = [p.ratio.cpu().numpy().copy() for p in rig.pipes]
    params = [list(p.params) for p in rig.pipes]
    infos = [p.info.cpu().numpy().copy() for p in rig.pipes]
    frames = rig.parts[0].cpu().numpy().copy()
    # the S sub-batches hold the same frames: every output equal to sub-batch 0's
    assert bench.outputs_identical(rig)
    for i in range(1, S):
        assert np.array_equal(outs[i], outs[0]), i
        assert np.array_equal(ratios[i], ratios[0]), i
        assert params[i] == params[0], i
        assert np.array_equal(infos[i], infos[0]), i
    # the device's own dehaze output for these frames (a separate pipe, the un-chained call)
    probe = FramePipe(0, Fs, H, W, guard_s=bench.GUARD_S)
    probe.stage_dehaze(rig.parts[0])
    torch.cuda.synchronize()
    dehazed = probe.work.cpu().numpy().copy()
    probe.close()
    rig.close()
    exp = []
    for f in range(Fs):
        o, _ = orc.dehaze(frames[f], 15, full=True, guard_s=True)
        d = np.abs(dehazed[f].astype(int) - o.astype(int))
        assert d.max() <= 1 and (d != 0).mean() <= 1e-2, (f, d.max(), (d != 0).mean())
        st, _ = orc.histretch(dehazed[f], "RGB")
        v = orc.bgr_to_v(st)
        bs, cl = knee_mirror.select_parameters(orc.sweep(orc.gaussian3(v)))
        assert params[0][f] == (bs, cl), (f, params[0][f], (bs, cl))
        e = orc.hsv_replace_v(st, orc.clahe(v, float(cl), bs, bs))
        assert np.array_equal(outs[0][f], e), f
        exp.append(e)
    # after the second step: frame 0 was matched against the previous step's last frame, frame 1 against frame 0
    er0, _, _ = orc.calcOverlap(exp[Fs - 1], exp[0], W, H, seed=1)
    er1, _, _ = orc.calcOverlap(exp[0], exp[1], W, H, seed=1)
    assert abs(ratios[0][0] - er0) <= 1e-6 and abs(ratios[0][1] - er1) <= 1e-6, (ratios[0], er0, er1)


def _random_desc(rng, n):
    d = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    d[:, 60] &= 0x3f
    d[:, 61:] = 0                                   # 486 payload bits, padded to 512
    return d


def _upload(ctx, f, slot, desc):
    n = len(desc)
    kps = np.zeros(max(n, 1), vs.KP_DTYPE)
    kps["x"] = np.linspace(8, 632, max(n, 1)); kps["y"] = np.linspace(8, 352, max(n, 1))
    ctx.call("uwip_features_upload", f._h, slot, 360, 640, C.c_void_p(kps.ctypes.data), C.c_void_p(desc.ctypes.data), n)


@pytest.mark.parametrize("form", ["4", "3"])
def test_matcher_config4_2048x2048_exact(ctx, orc, form, monkeypatch):
    """The workload of roofline.matcher.config4_2048x2048: 64 pairs of full descriptor sets in ONE launch -- with the FP4
    operands of round 5 (UWIP_MATCH_FORM=4, the default: v_mfma_scale_f32_16x16x128_f8f6f4, E2M1 nibbles, float32 sums) and
    with round 4's i8 operands (=3): the same (index, distance) pairs, exact."""
    monkeypatch.setenv("UWIP_MATCH_FORM", form)          # re-read per call in a process started with UWIP_TEST_HOOKS=1
    rng = np.random.default_rng(7)
    pairs, K = 64, 2048
    f = vs.Features(ctx, 2 * pairs)
    descs = []
    for slot in range(2 * pairs):
        d = _random_desc(rng, K)
        # near-duplicates: make exact distance ties and small distances happen (random 486-bit strings sit near 243)
        if slot % 2 == 1:
            q = descs[slot - 1]
            d[:K // 4] = q[rng.permutation(K)[:K // 4]]
            flip = rng.integers(0, 60, K // 4)
            d[np.arange(K // 4), flip] ^= np.uint8(1) << rng.integers(0, 8, K // 4).astype(np.uint8)
        descs.append(d)
        _upload(ctx, f, slot, d)
    res = vs.match_pairs(ctx, f, f, [2 * i for i in range(pairs)], [2 * i + 1 for i in range(pairs)], 640, 480, seed=1,
                         want_matches=True)
    idx, dist = res["idx"].cpu().numpy(), res["dist"].cpu().numpy()
    for p in range(pairs):
        eidx, edist = orc.match_knn2(descs[2 * p], descs[2 * p + 1])
        assert np.array_equal(dist[p, :K], edist), p
        assert np.array_equal(idx[p, :K], eidx), p
    f.close()


@pytest.mark.parametrize("form", ["4", "3", "5", "6"])
@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2047), (63, 65), (65, 63), (255, 257), (257, 255), (2047, 1), (2047, 2048),
                                   (2048, 2047), (64, 64), (256, 2), (2, 256)])
def test_matcher_ragged_counts_exact(ctx, orc, nq, nt, form, monkeypatch):
    monkeypatch.setenv("UWIP_MATCH_FORM", form)          # 4 = FP4 (default), 3 = i8, 5 = FP4 128 columns per barrier, 6 = FP4 4-wave blocks
    rng = np.random.default_rng(nq * 4099 + nt)
    f = vs.Features(ctx, 2)
    # slots first filled to capacity, then refilled with the ragged counts: rows past the count hold stale descriptors
    for slot in (0, 1):
        _upload(ctx, f, slot, _random_desc(rng, 2048))
    dq, dt = _random_desc(rng, nq), _random_desc(rng, nt)
    _upload(ctx, f, 0, dq)
    _upload(ctx, f, 1, dt)
    res = vs.match_pairs(ctx, f, f, [0], [1], 640, 480, seed=1, want_matches=True)
    idx, dist = res["idx"].cpu().numpy()[0], res["dist"].cpu().numpy()[0]
    eidx, edist = orc.match_knn2(dq, dt)
    assert np.array_equal(idx[:nq], eidx[:nq]) and np.array_equal(dist[:nq], edist[:nq])
    if nt == 1:
        assert np.all(idx[:nq, 1] == -1)
    f.close()
