"""GPU: the whole pipe (bgdehaze -> histretch -> aclahe -> overlap) against the oracle chained the
same way, on a small stream.  The dehaze tail is ill-conditioned w.r.t. 1-ulp differences (see
test_dehaze_gpu), so the chain is re-synchronised after dehaze: the oracle continues from the
device's dehaze output, and every later stage must then match exactly."""
import os
import sys

import numpy as np
import pytest

import _knee_mirror as knee_mirror   # the scipy mirror of functions.py:49-93 (a checker: lives with the tests)
import torch

from uwimageproc_amd import aclahe, synth
from uwimageproc_amd.pipeline import FramePipe

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dehaze_oracle as dz  # noqa: E402

pytestmark = pytest.mark.gpu


def test_full_pipe_small_stream(orc):
    F, H, W = 3, 270, 480
    frames = synth.uw_stream(0, F, H, W)
    pipe = FramePipe(0, F, H, W, video_size=(640, 480), guard_s=True)
    src = torch.from_numpy(frames).cuda()
    pipe.stage_dehaze(src)
    torch.cuda.synchronize()
    dehazed = pipe.work.cpu().numpy().copy()
    for f in range(F):
        exp = dz.to_u8(dz.adaptiveExp_map(dz.normalize_input(frames[f]), 15, guard_s=True))
        diff = np.abs(dehazed[f].astype(int) - exp.astype(int))
        assert diff.max() <= 1 and (diff != 0).mean() <= 1e-2
    pipe.stage_histretch()
    pipe.stage_aclahe()
    pipe.stage_overlap()
    torch.cuda.synchronize()
    out = pipe.work.cpu().numpy()
    ratios = pipe.ratio.cpu().numpy()
    exp_frames = []
    for f in range(F):
        st, _ = orc.histretch(dehazed[f], "RGB")
        v = orc.bgr_to_v(st)
        bs, cl = knee_mirror.select_parameters(orc.sweep(orc.gaussian3(v)))
        assert pipe.params[f] == (bs, cl)
        e = orc.hsv_replace_v(st, orc.clahe(v, float(cl), bs, bs))
        assert np.array_equal(out[f], e), f
        exp_frames.append(e)
    for f in range(F):
        key = exp_frames[f - 1] if f > 0 else exp_frames[0]
        er, info, _ = orc.calcOverlap(key, exp_frames[f], 640, 480, seed=1)
        assert abs(ratios[f] - er) <= 1e-6, (f, ratios[f], er)
    # second batch: frame 0 is matched against the previous batch's last frame
    frames2 = synth.uw_stream(F, F, H, W)
    pipe.run(torch.from_numpy(frames2).cuda())
    torch.cuda.synchronize()
    out2 = pipe.work.cpu().numpy()
    er, _, _ = orc.calcOverlap(out[F - 1], out2[0], 640, 480, seed=1)
    assert abs(float(pipe.ratio.cpu()[0]) - er) <= 1e-6
    pipe.close()


@pytest.mark.parametrize("letters,full,fixed", [("RGB", True, False), ("B", True, False), ("RGBVR", True, False),
                                                ("VRG", True, False), ("RlG", True, True), ("RGB", False, False),
                                                ("", True, False), ("zz", True, False)])
def test_chained_dehaze_histretch_equals_two_calls(letters, full, fixed):
    """uwip_dehaze_histretch (histogram counted by the dehaze writer) == uwip_dehaze then uwip_histretch_ex, byte for
    byte: a BGR run first (histogram reused), a colour-space letter first (it must not be), several frames and a
    ragged width so the tail chunks of both paths are exercised."""
    import uwimageproc_amd as uw
    from uwimageproc_amd import bgdehaze as bg, preprocessing as pp
    ctx = uw.Context(0)
    frames = torch.from_numpy(synth.uw_stream(3, 3, 131, 203)).cuda()
    two = bg.dehaze(ctx, frames, 15, full=full, guard_s=True)
    pp.histretch(ctx, two, letters, 2, 98, fixed_order=fixed)
    one = bg.dehaze_histretch(ctx, frames, letters, 2, 98, 15, full=full, guard_s=True, fixed_order=fixed)
    assert torch.equal(one, two)
    # again on the same context: the histogram workspace must be re-zeroed, not accumulated
    again = bg.dehaze_histretch(ctx, frames, letters, 2, 98, 15, full=full, guard_s=True, fixed_order=fixed)
    assert torch.equal(again, two)


def test_config3_dehaze_histretch_4k(orc):
    """BASELINE config 3: bgdehaze -> histretch chained on a 3840x2160 frame.  Full-size checks through
    size-independent properties (determinism, min-max span, border transmission) plus an exact
    oracle comparison of the histretch stage on the device's own dehaze output."""
    import ctypes as C
    from uwimageproc_amd import batch_of, bgdehaze as bg, preprocessing as pp
    import uwimageproc_amd as uw
    ctx = uw.Context(0)
    img = synth.uw_stream(0, 1, 2160, 3840)[0]
    t = torch.from_numpy(img).cuda()
    a = bg.dehaze(ctx, t, 15, full=True, guard_s=True)
    b = bg.dehaze(ctx, t, 15, full=True, guard_s=True)
    assert torch.equal(a, b)
    ah = a.cpu().numpy()
    assert ah.min() == 0 and ah.max() == 255
    pp.histretch(ctx, a, "RGB")
    exp, _ = orc.histretch(ah, "RGB")
    assert np.array_equal(a.cpu().numpy(), exp)
    ctx.close()


def test_config5_full_pipe_4k(orc):
    """BASELINE config 5: the full pipe on 3840x2160 frames.  The oracle continues from the device's own
    dehaze output (see the module docstring); every later stage must match it exactly, and a batch of two
    must equal the frames run one at a time."""
    F, H, W = 2, 2160, 3840
    frames = synth.uw_stream(0, F, H, W)
    pipe = FramePipe(0, F, H, W, guard_s=True)
    src = torch.from_numpy(frames).cuda()
    pipe.stage_dehaze(src)
    torch.cuda.synchronize()
    dehazed = pipe.work.cpu().numpy().copy()
    pipe.stage_histretch()
    pipe.stage_aclahe()
    pipe.stage_overlap()
    torch.cuda.synchronize()
    out = pipe.work.cpu().numpy().copy()
    ratios = pipe.ratio.cpu().numpy().copy()
    params = list(pipe.params)
    exp = []
    for f in range(F):
        st, _ = orc.histretch(dehazed[f], "RGB")
        v = orc.bgr_to_v(st)
        if f == 0:   # the 255-evaluation sweep of the oracle takes a while at this size: one frame
            assert params[f] == knee_mirror.select_parameters(orc.sweep(orc.gaussian3(v)))
        bs, cl = params[f]
        e = orc.hsv_replace_v(st, orc.clahe(v, float(cl), bs, bs))
        assert np.array_equal(out[f], e), f
        exp.append(e)
    er, _, _ = orc.calcOverlap(exp[0], exp[1], W, H, seed=1)
    assert abs(ratios[1] - er) <= 1e-6
    pipe.close()
    # one frame at a time
    single = FramePipe(0, 1, H, W, guard_s=True)
    for f in range(F):
        o, r = single.run(src[f:f + 1])
        torch.cuda.synchronize()
        assert np.array_equal(o.cpu().numpy()[0], out[f])
        assert single.params[0] == params[f]
        if f > 0:
            assert abs(float(r.cpu()[0]) - ratios[f]) <= 1e-6
    single.close()


@pytest.mark.parametrize("kind", ["constant", "black", "white", "two_level", "one_pixel"])
def test_degenerate_frames_run_clean(kind):
    """Frames a real video starts or ends with.  A constant frame normalises to 0/0 (NaN) in bgdehaze -- the
    reference then writes whatever (NaN * 255).astype(uint8) gives -- so there is nothing to compare against; the
    pipe must simply complete, deterministically, with a finite 8-bit result and the documented -2.0 / -1.0 overlap
    codes or a ratio."""
    H, W = 96, 128
    base = {"constant": np.full((H, W, 3), 77, np.uint8), "black": np.zeros((H, W, 3), np.uint8),
            "white": np.full((H, W, 3), 255, np.uint8)}
    if kind in base:
        img = base[kind]
    elif kind == "two_level":
        img = np.where((np.indices((H, W)).sum(0) // 8 % 2)[..., None] == 0, 30, 220).astype(np.uint8).repeat(3, axis=2)
    else:
        img = np.full((H, W, 3), 10, np.uint8)
        img[40, 50] = (255, 200, 100)
    frames = np.stack([img, img])
    outs = []
    for _ in range(2):
        pipe = FramePipe(0, 2, H, W, guard_s=True)
        o, r = pipe.run(torch.from_numpy(frames).cuda())
        torch.cuda.synchronize()
        outs.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), list(pipe.params)))
        pipe.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][2] == outs[1][2]
    assert np.array_equal(outs[0][1], outs[1][1], equal_nan=True)
    assert np.array_equal(outs[0][0][0], outs[0][0][1])          # identical frames -> identical outputs


def test_context_shares_torchs_default_stream():
    """The handle of torch's default stream is 0; a context given that handle must run ON that stream (round 1
    silently made a private stream instead, so a producer enqueued by torch just before a library call was not
    ordered before it -- the stream aliasing behind the host-buffer bench variant's unordered copies)."""
    import ctypes as C
    import uwimageproc_amd as uw
    from uwimageproc_amd import batch_of
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    assert torch.cuda.current_stream(dev).cuda_stream == 0
    c = uw.Context(0, stream=0)
    img = torch.zeros((4, 1080, 1920, 3), dtype=torch.uint8, device=dev)
    fill = torch.full_like(img, 7)
    hist = torch.zeros((4, 3, 256), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    torch.cuda._sleep(400_000_000)          # ~0.2 s of GPU time on torch's current (default) stream
    img.copy_(fill)                         # the producer, queued behind the sleep
    b = batch_of(img)
    c.call("uwip_getHistogram", C.byref(b), C.c_void_p(hist.data_ptr()))   # must be ordered after it
    torch.cuda.synchronize()
    h = hist.cpu().numpy()
    assert (h[:, :, 7] == 1080 * 1920).all() and h.sum() == 4 * 3 * 1080 * 1920
    c.close()
    # a context created without a stream owns a private one and is NOT ordered with torch (by design)
    c2 = uw.Context(0)
    c2.close()


def test_host_buffer_front_end_equals_device_path():
    """upload -> pipe -> download through the library's page-locked buffers and stream-ordered copies gives the
    bytes the HBM-resident path gives; two pipes on two streams from two host threads (the bench's arrangement)."""
    import threading
    F, H, W = 2, 270, 480
    dev = torch.device("cuda", 0)
    frames = [synth.uw_stream(10 * i, F, H, W) for i in range(2)]
    ref = []
    for i in range(2):
        p = FramePipe(0, F, H, W, video_size=(640, 480), guard_s=True)
        out, ratio = p.run(torch.from_numpy(frames[i]).cuda())
        torch.cuda.synchronize()
        ref.append((out.cpu().numpy().copy(), ratio.cpu().numpy().copy()))
        p.close()
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    pipes, bufs = [], []
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            pipes.append(FramePipe(0, F, H, W, video_size=(640, 480), guard_s=True))
        bufs.append(pipes[i].host_buffers())
        bufs[i][0][...] = frames[i]
        bufs[i][1][...] = 0

    def work(i):
        for _ in range(2):
            pipes[i].have_prev = False
            pipes[i].run_host(*bufs[i])
        pipes[i].sync()
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        assert np.array_equal(bufs[i][1], ref[i][0])
        assert np.allclose(pipes[i].ratio.cpu().numpy(), ref[i][1], atol=0)
        for a in bufs[i]:
            pipes[i].ctx.host_free(a)
        pipes[i].close()


def test_host_buffer_front_end_prefetched_batches():
    """A stream of different batches through run_host with the next batch prefetched (upload and download on their own
    streams, two source and two result buffers): every downloaded batch equals the HBM-resident path's result, and the
    overlap ratios carry across the batches as they do there."""
    F, H, W, NB = 2, 270, 480, 5
    batches = [synth.uw_stream(F * b, F, H, W) for b in range(NB)]
    ref_pipe = FramePipe(0, F, H, W, video_size=(640, 480), guard_s=True)
    ref = []
    for b in range(NB):
        out, ratio = ref_pipe.run(torch.from_numpy(batches[b]).cuda())
        torch.cuda.synchronize()
        ref.append((out.cpu().numpy().copy(), ratio.cpu().numpy().copy()))
    ref_pipe.close()
    with torch.cuda.stream(torch.cuda.Stream()):
        pipe = FramePipe(0, F, H, W, video_size=(640, 480), guard_s=True)
    h_in = [pipe.ctx.host_alloc((F, H, W, 3)) for _ in range(NB)]
    h_out = [pipe.ctx.host_alloc((F, H, W, 3)) for _ in range(NB)]
    for b in range(NB):
        h_in[b][...] = batches[b]
        h_out[b][...] = 0
    ratios = []
    for b in range(NB):
        # batch 3 arrives unannounced (no prefetch by batch 2): the pipe must upload it itself
        nxt = h_in[b + 1] if b + 1 < NB and b != 2 else None
        pipe.run_host(h_in[b], h_out[b], prefetch=nxt)
        with torch.cuda.stream(pipe.stream):
            ratios.append(pipe.ratio.clone())      # ordered behind this batch's kernels on the pipe's stream
    pipe.sync()
    for b in range(NB):
        assert np.array_equal(h_out[b], ref[b][0]), b
        assert np.array_equal(ratios[b].cpu().numpy(), ref[b][1]), b
    for a in h_in + h_out:
        pipe.ctx.host_free(a)
    pipe.close()


def test_stream_driver_seam_rule_with_the_real_pipe():
    """uwimageproc_amd.stream: two ranks' slices (run one after the other here, each with its own FramePipe) give frame
    for frame the output, ratio and CLAHE parameters of the single-rank stream -- the one-frame halo makes the overlap
    of a slice's first frame exact."""
    from uwimageproc_amd import stream
    n, B, H, W = 7, 3, 216, 384
    frames = synth.uw_stream(0, n, H, W)
    def run(rank, world):
        pipe = FramePipe(0, B, H, W, video_size=(640, 480), guard_s=True)
        drv = stream.StreamDriver(n, rank, world, B, stream.pipe_process(pipe))
        outs = {}
        r, p = drv.run(lambda i: frames[i], sink=lambda i, f: outs.__setitem__(i, f.copy()))
        pipe.close()
        return drv, outs, r, p
    _, o1, r1, p1 = run(0, 1)
    got_o, got_r, got_p = {}, [], []
    for rank in range(2):
        drv, o, r, p = run(rank, 2)
        assert sorted(o) == list(range(drv.start, drv.stop))
        got_o.update(o); got_r += r; got_p += p
    assert got_p == p1 and np.array_equal(np.array(got_r, np.float32), np.array(r1, np.float32))
    for i in range(n):
        assert np.array_equal(got_o[i], o1[i]), i


@pytest.mark.parametrize("shape", [(270, 480), (275, 483)])
def test_pipe_c_abi_reference_defaults_step_and_host_form(orc, shape):
    """uwip_pipe_* driven through ctypes alone (what a C / C++ integrator binds, include/uwip.h "the whole per-frame chain"):
    with uwip_pipe_config_default -- the reference's rules: S unguarded, >= 4 good matches, fixed detector threshold -- two
    steps through uwip_pipe_step equal the chain made by hand from the stage entry points (uwip_dehaze_histretch, uwip_bgr_to_v,
    uwip_aclahe_auto_ex, uwip_hsv_replace_v, uwip_overlap_detect / _match with the feature-slot carry), and the same two
    batches through uwip_pipe_step_host (frames AND ratios downloaded, tickets) give the same bytes."""
    import ctypes as C
    from uwimageproc_amd import Context, PipeConfig, batch_of
    F, (H, W) = 3, shape              # 275 x 483: rows of 1449 bytes -- every kernel of the chain on its unaligned / scalar path
    ctx = Context(0)
    l = ctx._l
    cfg = PipeConfig()
    assert l.uwip_pipe_config_default(C.byref(cfg), F, H, W) == 0
    cfg.videoWidth, cfg.videoHeight = 640, 480
    h = C.c_void_p()
    ctx.call("uwip_pipe_create", C.byref(cfg), None, C.byref(h))

    def pc(name, *a):
        rc = getattr(l, name)(h, *a)
        assert rc == 0, (name, rc, l.uwip_pipe_last_error(h))
    batches = [torch.from_numpy(synth.uw_stream(F * b, F, H, W)).cuda() for b in range(2)]
    outs = [torch.empty_like(batches[0]) for _ in range(2)]
    ratios = [torch.empty(F, dtype=torch.float32, device="cuda") for _ in range(2)]
    infos = [torch.zeros((F, 8), dtype=torch.int32, device="cuda") for _ in range(2)]
    params = []
    for b in range(2):
        ib, ob = batch_of(batches[b]), batch_of(outs[b])
        pc("uwip_pipe_step", C.byref(ib), C.byref(ob), C.c_void_p(ratios[b].data_ptr()), C.c_void_p(infos[b].data_ptr()))
        bs, cl = (C.c_int32 * F)(), (C.c_int32 * F)()
        pc("uwip_pipe_last_params", bs, cl)
        params.append(list(zip(bs, cl)))
    pc("uwip_pipe_sync")
    # in place dehaze is refused, a wrong geometry too
    ib = batch_of(batches[0])
    assert l.uwip_pipe_step(h, C.byref(ib), C.byref(ib), C.c_void_p(ratios[0].data_ptr()), None) != 0
    assert b"distinct" in l.uwip_pipe_last_error(h)
    small = batch_of(batches[0][:2])
    assert l.uwip_pipe_step(h, C.byref(small), C.byref(batch_of(outs[0][:2])), C.c_void_p(ratios[0].data_ptr()), None) != 0

    # the same chain by hand
    c2 = Context(0)
    fh = C.c_void_p()
    c2.call("uwip_features_create", F + 1, C.byref(fh))
    v, vo = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"), torch.empty((F, H, W), dtype=torch.uint8, device="cuda")
    pq, pt = (C.c_int32 * F)(*range(1, F + 1)), (C.c_int32 * F)(*range(F))
    for b in range(2):
        o = torch.empty_like(batches[b])
        r = torch.empty(F, dtype=torch.float32, device="cuda")
        ib, ob, vb, vob = batch_of(batches[b]), batch_of(o), batch_of(v), batch_of(vo)
        c2.call("uwip_dehaze_histretch", C.byref(ib), C.byref(ob), 15, 1, b"RGB", 2, 98, 0)         # UWIP_DEHAZE_FULL, unguarded
        c2.call("uwip_bgr_to_v", C.byref(ob), C.byref(vb))
        bs, cl = (C.c_int32 * F)(), (C.c_int32 * F)()
        c2.call("uwip_aclahe_auto_ex", C.byref(vb), C.byref(vob), 0, 1, bs, cl)                     # PREFILTER, synchronous form
        c2.call("uwip_hsv_replace_v", C.byref(ob), C.byref(vob), C.byref(ob))
        if b == 0:
            c2.call("uwip_overlap_detect", C.byref(batch_of(o[0:1])), fh, 0)
        else:
            c2.call("uwip_features_copy", fh, F, fh, 0)
        c2.call("uwip_overlap_detect", C.byref(ob), fh, 1)
        c2.call("uwip_overlap_match", fh, fh, pq, pt, F, 640, 480, 1, C.c_void_p(r.data_ptr()), None, None, None, None)
        c2.sync()
        assert torch.equal(o, outs[b]) and torch.equal(r, ratios[b]), b
        assert params[b] == list(zip(bs, cl)), b
    # frame 1 against frame 0 by the oracle, the reference's rule: the ratio of the device's own frames
    o0 = outs[0].cpu().numpy()
    er, _, _ = orc.calcOverlap(o0[0], o0[1], 640, 480, seed=1)
    assert abs(float(ratios[0].cpu()[1]) - er) <= 1e-6
    c2._l.uwip_features_destroy(fh)
    c2.close()

    # host-buffer form on a fresh pipe: same bytes, ratios downloaded too
    h2 = C.c_void_p()
    ctx.call("uwip_pipe_create", C.byref(cfg), None, C.byref(h2))
    h_in = [ctx.host_alloc((F, H, W, 3)) for _ in range(2)]
    h_out = [ctx.host_alloc((F, H, W, 3)) for _ in range(2)]
    h_r = [ctx.host_alloc((F,), "float32") for _ in range(2)]
    for b in range(2):
        h_in[b][...] = batches[b].cpu().numpy()
    t = (C.c_uint64 * 3)()
    for b in range(2):
        rc = l.uwip_pipe_step_host(h2, C.c_void_p(h_in[b].ctypes.data), C.c_void_p(h_out[b].ctypes.data), C.c_void_p(h_r[b].ctypes.data),
                                   C.c_void_p(h_in[1].ctypes.data) if b == 0 else None, t)
        assert rc == 0, l.uwip_pipe_last_error(h2)
        assert t[0] and t[1] and t[2]
        assert l.uwip_pipe_wait(h2, t[1]) == 0 and l.uwip_pipe_wait(h2, t[2]) == 0
        assert np.array_equal(h_out[b], outs[b].cpu().numpy()), b
        assert np.array_equal(h_r[b], ratios[b].cpu().numpy()), b
    assert l.uwip_pipe_sync(h2) == 0
    dv, df, dr = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert l.uwip_pipe_device_results(h2, C.byref(dv), C.byref(df), C.byref(dr), None) == 0 and dv.value and df.value and dr.value
    for a in h_in + h_out + h_r:
        ctx.host_free(a)
    assert l.uwip_pipe_destroy(h2) == 0 and l.uwip_pipe_destroy(h) == 0
    ctx.close()
