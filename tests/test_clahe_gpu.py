"""GPU parity: CLAHE / entropy / sweep kernels (through the C ABI) vs the CPU
oracle.  Integer stages and the float32 interpolation are bit-exact; entropy
is compared within 1e-5 abs (device log2 vs libm)."""
import numpy as np
import pytest

import _knee_mirror as knee_mirror   # the scipy mirror of functions.py:49-93 (a checker: lives with the tests)
import torch

from uwimageproc_amd import aclahe, synth

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _v(orc, idx, rows, cols):
    return orc.bgr_to_v(synth.uw_frame(idx, rows, cols))


def test_bgr_to_v(ctx, orc):
    for shape in ((37, 53), (64, 64), (1080, 1920)):
        img = synth.uw_frame(2, *shape)
        v = aclahe.bgr_to_v(ctx, _dev(img)).cpu().numpy()
        assert np.array_equal(v, orc.bgr_to_v(img))
    batch = synth.uw_batch(3, 3, 40, 48)
    v = aclahe.bgr_to_v(ctx, _dev(batch)).cpu().numpy()
    for f in range(3):
        assert np.array_equal(v[f], orc.bgr_to_v(batch[f]))


@pytest.mark.parametrize("shape,grid,clip,rule", [
    ((64, 96), (2, 2), 0.0, 0), ((64, 96), (4, 4), 2.0, 0), ((30, 50), (4, 4), 3.5, 0),
    ((37, 53), (8, 8), 40.0, 0), ((37, 53), (8, 8), 1.0, 1), ((24, 24), (3, 5), 4.0, 0),
    ((270, 480), (8, 8), 2.5, 0), ((270, 480), (16, 16), 12.0, 0), ((270, 480), (32, 32), 25.0, 0),
    ((135, 241), (32, 32), 0.5, 0), ((1080, 1920), (8, 8), 3.0, 0), ((1080, 1920), (16, 16), 2.0, 0),
    ((1080, 1920), (32, 32), 7.5, 1),
])
def test_clahe_bit_exact(ctx, orc, shape, grid, clip, rule):
    src = _v(orc, 5, *shape)
    exp, exp_luts = orc.clahe(src, clip, grid[0], grid[1], rule, want_luts=True)
    c = aclahe.CLAHE(ctx, clip, grid, rule)
    t = _dev(src)
    luts = c.luts(t).cpu().numpy()[0]
    assert np.array_equal(luts, exp_luts)
    out = c.apply(t).cpu().numpy()
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("shape,grid", [((300, 517), (4, 4)), ((500, 100), (6, 2)), ((280, 16), (1, 1)), ((520, 70), (3, 2)),
                                        ((1080, 1920), (16, 16)), ((1080, 1920), (32, 32)), ((2160, 3840), (32, 32)),
                                        ((543, 961), (7, 5)), ((64, 4100), (3, 1))])
def test_tilehist_general_slot_keyed_form(ctx, orc, shape, grid):
    """k_clahe_tilehist<2> (round 4): tiles of >= 4096 pixels that are padded, unaligned or not a multiple of 16 wide --
    tail units loaded ending at the tile's last in-image column (a last tile narrower than 16 in-image pixels reads its
    left neighbour's pixels and masks them), an image exactly 16 columns wide, reflect-101 padding rows / columns, a
    strided view at an odd address, several frames.  LUTs (i.e. the histograms) and the output bit-exact."""
    rows, cols = shape
    rng = np.random.default_rng(rows * 7 + cols)
    frames = rng.integers(0, 256, (2, rows, cols), dtype=np.uint8)
    frames[1] = _v(orc, 9, rows, cols) if cols >= 8 else frames[1]
    c = aclahe.CLAHE(ctx, 2.5, grid)
    t = _dev(frames)
    luts = c.luts(t).cpu().numpy()
    out = c.apply(t).cpu().numpy()
    for f in range(2):
        exp, exp_luts = orc.clahe(frames[f], 2.5, grid[0], grid[1], 0, want_luts=True)
        assert np.array_equal(luts[f], exp_luts), f
        assert np.array_equal(out[f], exp), f
    if rows * cols < 1 << 20:
        step = cols + 13
        buf = torch.zeros(rows * step + 64, dtype=torch.uint8, device="cuda")
        view = buf[5:5 + rows * step].view(rows, step)[:, :cols]
        view.copy_(t[0])
        assert np.array_equal(c.luts(view).cpu().numpy()[0], orc.clahe(frames[0], 2.5, grid[0], grid[1], 0, want_luts=True)[1])


@pytest.mark.parametrize("shape,grid", [((40, 20), (32, 32)), ((33, 32), (32, 32)), ((64, 27), (32, 8)), ((543, 1915), (49, 5)),
                                        ((200, 1915), (49, 3)), ((97, 1000), (62, 4)), ((97, 630), (33, 3))])
def test_band_geometries_one_pixel_tiles_and_padding_over_several_parts(ctx, orc, shape, grid):
    """ADVICE r4 (k_clahe_band): tiles ONE pixel wide (16 <= cols <= 32 under the 32 x 32 grid every aclahe sweep runs: the
    multiply-high division does not exist for tw = 1 -> the separate kernels), and reflect-101 padding columns that cover
    more tiles than the last part of a row of tiles owns (cols = 1915, gx = 49 -> tw = 40, 16 tiles per block, last part =
    tile 48 alone, padding over tiles 47 and 48: every part now counts the padding of its own tiles).  LUTs, output and --
    for the first two -- the whole sweep table's histograms against the oracle."""
    rows, cols = shape
    rng = np.random.default_rng(rows * 11 + cols)
    frames = rng.integers(0, 256, (2, rows, cols), dtype=np.uint8)
    frames[1] = _v(orc, 11, rows, cols)
    c = aclahe.CLAHE(ctx, 3.0, grid)
    t = _dev(frames)
    luts = c.luts(t).cpu().numpy()
    out = c.apply(t).cpu().numpy()
    for f in range(2):
        exp, exp_luts = orc.clahe(frames[f], 3.0, grid[0], grid[1], 0, want_luts=True)
        assert np.array_equal(luts[f], exp_luts), f
        assert np.array_equal(out[f], exp), f
    if cols <= 32:
        tab = aclahe.sweep(ctx, t).cpu().numpy()
        for f in range(2):
            assert np.abs(tab[f] - orc.sweep(frames[f])).max() <= 1e-5, f


@pytest.mark.parametrize("kind", ["random", "constant", "two_level", "ramp"])
def test_clahe_adversarial(ctx, orc, kind):
    src = np.ascontiguousarray(synth.adversarial(kind, 100, 140)[..., 0])
    for g, cl in ((4, 2.0), (8, 0.0), (16, 40.0)):
        exp = orc.clahe(src, cl, g, g)
        out = aclahe.CLAHE(ctx, cl, (g, g)).apply(_dev(src)).cpu().numpy()
        assert np.array_equal(out, exp), (kind, g, cl)


def test_clahe_batch_and_per_frame(ctx, orc):
    frames = np.stack([_v(orc, 40 + i, 120, 200) for i in range(6)])
    t = _dev(frames)
    out = aclahe.CLAHE(ctx, 3.0, (8, 8)).apply(t).cpu().numpy()
    for f in range(6):
        assert np.array_equal(out[f], orc.clahe(frames[f], 3.0, 8, 8))
    cls = [1.0, 7.5, 0.0, 24.5, 3.0, 7.5]
    grids = [8, 2, 32, 8, 16, 2]
    out = aclahe.clahe_per_frame(ctx, t, cls, grids).cpu().numpy()
    for f in range(6):
        assert np.array_equal(out[f], orc.clahe(frames[f], cls[f], grids[f], grids[f])), f


def test_clahe_strided_unaligned(ctx, orc):
    rows, cols = 45, 77
    src = _v(orc, 6, rows, cols)
    exp = orc.clahe(src, 2.0, 4, 4)
    step = cols + 11
    buf = torch.zeros(rows * step + 8, dtype=torch.uint8, device="cuda")
    view = buf[3:3 + rows * step].view(rows, step)[:, :cols]
    view.copy_(_dev(src))
    dst = torch.zeros_like(view)
    aclahe.CLAHE(ctx, 2.0, (4, 4)).apply(view, dst)
    assert np.array_equal(dst.cpu().numpy(), exp)


def test_entropy(ctx, orc):
    frames = np.stack([_v(orc, 50 + i, 90, 160) for i in range(4)])
    e = aclahe.aclaheEntropy(ctx, _dev(frames)).cpu().numpy()
    for f in range(4):
        assert abs(float(e[f]) - orc.entropy(frames[f])) <= 1e-5


@pytest.mark.parametrize("shape", [(64, 96), (135, 240), (270, 480)])
def test_sweep_vs_oracle(ctx, orc, shape):
    frames = np.stack([_v(orc, 60 + i, *shape) for i in range(2)])
    tab = aclahe.sweep(ctx, _dev(frames)).cpu().numpy()
    assert tab.shape == (2, 5, 51)
    for f in range(2):
        exp = orc.sweep(frames[f])
        assert np.abs(tab[f] - exp).max() <= 1e-5, np.abs(tab[f] - exp).max()


def _mixed_plane(seed, rows, cols):
    """Regions whose tile histograms stop being clipped at different clip limits: 6 grey levels (every limit clips),
    ~20 levels (only the limits of the last group repeat), ~40 levels, white noise (nothing clips beyond the first few
    limits), a constant patch, blown-out (255) areas; region borders do not follow any tile grid."""
    rng = np.random.default_rng(seed)
    v = np.empty((rows, cols), np.uint8)
    xs = [0, cols * 3 // 11, cols * 5 // 11, cols * 8 // 11, cols]
    v[:, xs[0]:xs[1]] = rng.integers(0, 6, (rows, xs[1] - xs[0])) * 40 + 10
    v[:, xs[1]:xs[2]] = rng.integers(0, 20, (rows, xs[2] - xs[1])) * 12 + 3
    v[:, xs[2]:xs[3]] = rng.integers(0, 256, (rows, xs[3] - xs[2]))
    v[:, xs[3]:xs[4]] = rng.integers(0, 40, (rows, xs[4] - xs[3])) * 6
    v[rows // 3: rows // 2, cols // 4: cols // 2] = 77
    v[rows // 2: (rows * 3) // 4, cols // 8: (cols * 7) // 8] = 255        # a blown-out band (the sweep's saturated-wave path)
    v[(rows * 3) // 4:, : cols // 3][::2] = 255                            # ... and one that never fills a whole wave
    return v


@pytest.mark.parametrize("shape", [(1, 2), (5, 7), (33, 65), (96, 160), (270, 480), (301, 517)])
def test_sweep_histograms_exact_vs_oracle(ctx, orc, shape):
    """Every one of the 255 output histograms, count for count: the kernel evaluates a clip limit only where its LUTs
    differ from the previous limit's and books the repeats through tail / group counters, so the check is on the
    counts themselves (an entropy tolerance would hide single pixels)."""
    frames = np.stack([_mixed_plane(5, *shape), _v(orc, 61, *shape)])
    ent, hist = aclahe.sweep_histograms(ctx, _dev(frames))
    hist = hist.cpu().numpy()
    assert torch.equal(ent, aclahe.sweep(ctx, _dev(frames)))
    for f in range(2):
        for gi, g in enumerate(aclahe.BLOCK_SIZES):
            for ci, cl in enumerate(aclahe.CLIP_LIMITS):
                exp = np.bincount(orc.clahe(frames[f], float(cl), g, g).ravel(), minlength=256)
                assert np.array_equal(hist[f, gi, ci], exp), (f, g, cl)


def test_sweep_histograms_exact_1080p(ctx, orc):
    """Full size: the tap against the histogram of this library's own (oracle-exact) CLAHE apply, for limits of all three
    clip-limit groups on every grid, on a bench-like frame and on the mixed-region plane."""
    for src in (_v(orc, 70, 1080, 1920), _mixed_plane(9, 1080, 1920)):
        t = _dev(src)
        _, hist = aclahe.sweep_histograms(ctx, t)
        assert int(hist.sum(dim=-1).min()) == 1080 * 1920 and int(hist.sum(dim=-1).max()) == 1080 * 1920
        c = aclahe.CLAHE(ctx)
        for gi, g in enumerate(aclahe.BLOCK_SIZES):
            c.setTilesGridSize((g, g))
            for ci in (0, 1, 9, 16, 17, 25, 33, 34, 42, 50):
                c.setClipLimit(aclahe.CLIP_LIMITS[ci])
                exp = torch.bincount(c.apply(t).flatten().to(torch.int64), minlength=256)
                assert torch.equal(hist[0, gi, ci].to(torch.int64), exp), (g, ci)


def test_sweep_all_255_histograms_vs_oracle_1080p(ctx, orc):
    """BASELINE config 2's size, against the ORACLE itself (not this library's apply): all 5 x 51 output histograms of one
    1920x1080 bench-like frame, count for count (~255 oracle CLAHE applies, a few seconds of CPU)."""
    from concurrent.futures import ThreadPoolExecutor
    src = _v(orc, 70, 1080, 1920)
    _, hist = aclahe.sweep_histograms(ctx, _dev(src))
    hist = hist.cpu().numpy()[0]
    jobs = [(gi, ci) for gi in range(5) for ci in range(51)]

    def one(j):
        gi, ci = j
        g = aclahe.BLOCK_SIZES[gi]
        return np.bincount(orc.clahe(src, float(aclahe.CLIP_LIMITS[ci]), g, g).ravel(), minlength=256)
    with ThreadPoolExecutor(max_workers=8) as ex:
        exp = list(ex.map(one, jobs))
    for (gi, ci), e in zip(jobs, exp):
        assert np.array_equal(hist[gi, ci], e), (aclahe.BLOCK_SIZES[gi], aclahe.CLIP_LIMITS[ci])


def test_aclahe_auto_ex_vs_oracle_1080p(ctx, orc):
    """uwip_aclahe_auto_ex end to end at 1920x1080 (BASELINE config 2), both forms of the stage: the parameters equal the
    choice made on the oracle's own sweep table of the (pre-filtered) plane, the output equals the oracle's CLAHE of the
    unfiltered plane with them."""
    frames = synth.uw_stream(11, 2, 1080, 1920)
    v = np.stack([orc.bgr_to_v(f) for f in frames])
    t = torch.from_numpy(v).cuda()
    for prefilter in (True, False):
        dst, params = aclahe.auto(ctx, t, prefilter=prefilter)
        for f in range(2):
            src = orc.gaussian3(v[f]) if prefilter else v[f]
            bs, cl = knee_mirror.select_parameters(orc.sweep(src))
            assert params[f] == (bs, cl), (prefilter, f, params[f], (bs, cl))
            assert np.array_equal(dst[f].cpu().numpy(), orc.clahe(v[f], float(cl), bs, bs))


def test_sweep_histograms_exact_4k(ctx, orc):
    """3840x2160 (configs 3 / 5): the same check on one frame; at this size the 16x16 and 32x32 grids hold several cells
    per block and the 2x2 grid's cells are cut into many row chunks."""
    t = _dev(_mixed_plane(11, 2160, 3840))
    _, hist = aclahe.sweep_histograms(ctx, t)
    assert int(hist.sum(dim=-1).min()) == 2160 * 3840 and int(hist.sum(dim=-1).max()) == 2160 * 3840
    c = aclahe.CLAHE(ctx)
    for gi, g in enumerate(aclahe.BLOCK_SIZES):
        c.setTilesGridSize((g, g))
        for ci in (0, 3, 16, 17, 30, 34, 50):
            c.setClipLimit(aclahe.CLIP_LIMITS[ci])
            exp = torch.bincount(c.apply(t).flatten().to(torch.int64), minlength=256)
            assert torch.equal(hist[0, gi, ci].to(torch.int64), exp), (g, ci)


def test_sweep_1080p_properties(ctx, orc):
    # full-size: the sweep's histograms must equal those of materialised CLAHE outputs
    # (checked through entropy of our own bit-exact CLAHE apply), and spot-check the oracle
    src = _v(orc, 70, 1080, 1920)
    t = _dev(src)
    tab = aclahe.sweep(ctx, t).cpu().numpy()[0]
    c = aclahe.CLAHE(ctx)
    for gi, g in enumerate(aclahe.BLOCK_SIZES):
        c.setTilesGridSize((g, g))
        for ci in (0, 5, 50):
            c.setClipLimit(aclahe.CLIP_LIMITS[ci])
            e = float(aclahe.aclaheEntropy(ctx, c.apply(t)).cpu()[0])
            assert abs(tab[gi, ci] - e) <= 1e-6, (g, ci)
    assert abs(tab[2, 6] - orc.entropy(orc.clahe(src, 3.0, 8, 8))) <= 1e-5


def test_hsv_replace_v_bit_exact(ctx, orc):
    import ctypes as C
    from uwimageproc_amd import batch_of
    for shape in ((37, 53), (270, 480)):
        img = synth.adversarial("random", *shape) if shape[0] == 37 else synth.uw_frame(9, *shape)
        vnew = orc.clahe(orc.bgr_to_v(img), 3.0, 4, 4)
        t, v = _dev(img), _dev(vnew)
        out = torch.empty_like(t)
        tb, vb, ob = batch_of(t), batch_of(v), batch_of(out)
        torch.cuda.synchronize()
        ctx.call("uwip_hsv_replace_v", C.byref(tb), C.byref(vb), C.byref(ob))
        ctx.sync()
        assert np.array_equal(out.cpu().numpy(), orc.hsv_replace_v(img, vnew))


def test_aclahe_auto_matches_staged_path(ctx, orc):
    import ctypes as C
    from uwimageproc_amd import batch_of
    frames = np.stack([_v(orc, 80 + i, 135, 240) for i in range(5)])
    t = _dev(frames)
    out = torch.empty_like(t)
    bs, cl = (C.c_int32 * 5)(), (C.c_int32 * 5)()
    tb, ob = batch_of(t), batch_of(out)
    torch.cuda.synchronize()
    ctx.call("uwip_aclahe_auto", C.byref(tb), C.byref(ob), 0, bs, cl)
    ctx.sync()
    got = out.cpu().numpy()
    for f in range(5):
        ebs, ecl = knee_mirror.select_parameters(orc.sweep(frames[f]))          # scipy mirror on the oracle's table
        assert (bs[f], cl[f]) == (ebs, ecl)
        assert np.array_equal(got[f], orc.clahe(frames[f], float(ecl), ebs, ebs))


@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (7, 1), (37, 53), (270, 480), (1080, 1920)])
@pytest.mark.parametrize("rule", [0, 1])
def test_gaussian_blur3_vs_oracle(ctx, orc, shape, rule):
    """cv2.GaussianBlur(img,(3,3),0), ACLAHE.py:15: bit-exact against the oracle (both rounding rules), batches, strides"""
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    a = rng.integers(0, 256, (3,) + shape, dtype=np.uint8)
    a[1] = (a[1] // 64) * 2                      # few levels with many exact /16 ties
    if shape[1] in (1, 3):                       # a [F, H, 1] tensor would read as H x W x C: one 2-D plane at a time
        got = np.stack([aclahe.GaussianBlur3(ctx, torch.from_numpy(a[f]).cuda(), rule).cpu().numpy() for f in range(3)])
    else:
        got = aclahe.GaussianBlur3(ctx, torch.from_numpy(a).cuda(), rule).cpu().numpy()
    for f in range(3):
        assert np.array_equal(got[f], orc.gaussian3(a[f], rule)), (f, shape, rule)


def test_aclahe_auto_python_and_cpp_driver_modes(ctx, orc):
    """uwip_aclahe_auto_ex: ParametrosACLAHE (search on the blurred plane, final CLAHE on the plane itself) and the C++
    driver's form (search on the plane) against the oracle driven the same way."""
    from uwimageproc_amd import synth
    frames = synth.uw_stream(3, 2, 216, 384)
    v = np.stack([orc.bgr_to_v(f) for f in frames])
    t = torch.from_numpy(v).cuda()
    for prefilter in (True, False):
        dst, params = aclahe.auto(ctx, t, prefilter=prefilter)
        for f in range(2):
            src = orc.gaussian3(v[f]) if prefilter else v[f]
            bs, cl = knee_mirror.select_parameters(orc.sweep(src))
            assert params[f] == (bs, cl), (prefilter, f, params[f], (bs, cl))
            assert np.array_equal(dst[f].cpu().numpy(), orc.clahe(v[f], float(cl), bs, bs))     # final apply: unfiltered plane
        assert aclahe.ParametrosACLAHE(ctx, t, prefilter=prefilter) == params


@pytest.mark.parametrize("value", [200, 201])
def test_tilehist_16bit_slots_do_not_overflow_on_constant_4k_tiles(ctx, orc, value):
    """the slot-keyed tile histogram counts in 16-bit halves.  With many frames in the batch the host gives
    every (tile, frame) a single wave; a constant 1920x1080 tile of a 4K frame then puts > 65536 pixels into one slot
    unless the part size is bounded (launch_tilehist: TH_BP_PART_MAX).  Even value = low half carries into the odd bin,
    odd value = high half wraps."""
    F, H, W = 520, 2160, 3840
    t = torch.full((F, H, W), value, dtype=torch.uint8, device="cuda")
    c = aclahe.CLAHE(ctx, 0.0, (2, 2))
    luts = c.luts(t).cpu().numpy()
    _, exp = orc.clahe(np.full((H, W), value, np.uint8), 0.0, 2, 2, 0, want_luts=True)
    for f in (0, 1, F // 2, F - 1):
        assert np.array_equal(luts[f], exp), f
    assert (luts == luts[0]).all()
    del t
    torch.cuda.empty_cache()


def _device_select(ctx, tab_dev):
    import ctypes as C
    F = tab_dev.shape[0]
    par = torch.zeros((F, 4), dtype=torch.int32, device="cuda")
    knee = torch.zeros((F, 5), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.call("uwip_aclahe_select_device", C.c_void_p(tab_dev.data_ptr()), F, C.c_void_p(par.data_ptr()), C.c_void_p(knee.data_ptr()))
    ctx.sync()
    return par.cpu().numpy(), knee.cpu().numpy()


def _host_select(tabs):
    import ctypes as C
    from uwimageproc_amd import _native as nat
    tabs = np.ascontiguousarray(tabs, np.float32)
    F = len(tabs)
    bs, cl, knee = (C.c_int32 * F)(), (C.c_int32 * F)(), (C.c_int32 * (5 * F))()
    assert nat.lib().uwip_aclahe_select(tabs.ctypes.data_as(C.POINTER(C.c_float)), F, bs, cl, knee) == 0
    return np.array(list(bs)), np.array(list(cl)), np.array(list(knee)).reshape(F, 5)


def test_device_parameter_choice_equals_reference_goldens(ctx):
    """uwip_aclahe_select_device (one wavefront per curve, lm_core.hpp) on the tables of tests/golden/aclahe_knee.npz: the
    knee indices the REFERENCE's own functions.py gave (19 tables incl. the curves curve_fit gives up on), CL and BS."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "aclahe_knee.npz"), allow_pickle=False)
    tabs = np.ascontiguousarray(g["tables"], np.float32)
    par, knee = _device_select(ctx, _dev(tabs))
    assert np.array_equal(knee, g["idx"])
    bs, cl, hk = _host_select(tabs)
    assert np.array_equal(hk, knee) and np.array_equal(par[:, 0], bs) and np.array_equal(par[:, 1], cl)
    assert np.array_equal(par[:, 2], (2 * cl > 50).astype(np.int32))


def test_device_parameter_choice_equals_host_on_1024_tables(ctx, orc):
    """the device form against the host form on >= 1000 tables of the kind the pipe produces -- 1024 frames
    of the bench's motion stream and of the plain stream (270 x 480: the table does not care about the frame size), swept
    on the device, blurred as the pipe does; all 5120 knee indices, CL, BS and the need-evaluation flag equal."""
    F, H, W = 64, 270, 480
    n_equal = 0
    for k in range(16):
        frames = synth.uw_stream_motion(k * 7, F, H, W, seed0=1234 + 97 * k) if k % 2 == 0 else synth.uw_stream(k * 64, F, H, W)
        v = aclahe.bgr_to_v(ctx, _dev(frames))
        if k % 4 < 2:
            v = aclahe.GaussianBlur3(ctx, v)
        tab = aclahe.sweep(ctx, v)                                   # [F, 5, 51] on the device
        par, knee = _device_select(ctx, tab)
        bs, cl, hk = _host_select(tab.cpu().numpy())
        assert np.array_equal(hk, knee), k
        assert np.array_equal(par[:, 0], bs) and np.array_equal(par[:, 1], cl), k
        assert np.array_equal(par[:, 2], (2 * cl > 50).astype(np.int32))
        n_equal += F
    assert n_equal == 1024


def test_aclahe_auto_device_and_host_choice_agree(ctx, orc):
    """uwip_aclahe_auto_ex with the device choice (the default from 5 frames up) and with UWIP_ACLAHE_HOST_SELECT: same
    parameters, same image."""
    import ctypes as C
    from uwimageproc_amd import batch_of
    frames = np.stack([_v(orc, 300 + i, 270, 480) for i in range(5)])
    t = _dev(frames)
    outs = []
    for flags in (1, 1 | 2):
        dst = torch.empty_like(t)
        bs, cl = (C.c_int32 * 5)(), (C.c_int32 * 5)()
        sb, db = batch_of(t), batch_of(dst)
        ctx.call("uwip_aclahe_auto_ex", C.byref(sb), C.byref(db), 0, flags, bs, cl)
        ctx.sync()
        outs.append((list(zip(bs, cl)), dst.cpu().numpy()))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])


def _auto(ctx, t, flags):
    import ctypes as C
    from uwimageproc_amd import batch_of
    F = t.shape[0]
    dst = torch.empty_like(t)
    sb, db = batch_of(t), batch_of(dst)
    if flags & 4:                                            # UWIP_ACLAHE_ASYNC: parameters fetched afterwards
        ctx.call("uwip_aclahe_auto_ex", C.byref(sb), C.byref(db), 0, flags, None, None)
        bs, cl = (C.c_int32 * F)(), (C.c_int32 * F)()
        ctx.call("uwip_aclahe_last_params", bs, cl, F)
    else:
        bs, cl = (C.c_int32 * F)(), (C.c_int32 * F)()
        ctx.call("uwip_aclahe_auto_ex", C.byref(sb), C.byref(db), 0, flags, bs, cl)
    ctx.sync()
    return list(zip(bs, cl)), dst.cpu().numpy()


@pytest.mark.parametrize("shape", [(270, 480), (1080, 1920)])
def test_aclahe_async_equals_synchronous_forms(ctx, orc, shape, monkeypatch):
    """UWIP_ACLAHE_ASYNC (the choice on the device, the final CLAHE launched from device-side parameters, no host wait)
    against the synchronous device-choice form and the host form: same parameters, same bytes.  The frames are made to
    choose different block sizes (contrast / texture vary), so several of the five predicated grids take part; then the
    out-of-grid path: a forced clip limit of 30 (UWIP_ACLAHE_TEST_FORCE_CL) sends every frame through the exact block-size
    search -- k_aclahe_exact_bs on the device against the host-ordered CLAHE + entropy runs."""
    rows, cols = shape
    rng = np.random.default_rng(5)
    F = 8
    frames = np.stack([_v(orc, 500 + i, rows, cols) for i in range(F)])
    frames[1] = rng.integers(0, 256, (rows, cols), dtype=np.uint8)                       # noise
    frames[2] = (frames[2] // 32) * 32                                                   # posterised
    yy, xx = np.mgrid[0:rows, 0:cols]
    frames[3] = ((xx * 255) // cols).astype(np.uint8)                                    # ramp
    frames[4] = np.where((xx // 40 + yy // 40) % 2 == 0, 60, 190).astype(np.uint8)       # checkerboard
    frames[5] = np.clip(128 + 60 * np.sin(xx / 17.0) * np.cos(yy / 11.0) + rng.normal(0, 6, (rows, cols)), 0, 255).astype(np.uint8)
    t = _dev(frames)
    p_sync, o_sync = _auto(ctx, t, 1)
    p_host, o_host = _auto(ctx, t, 1 | 2)
    p_async, o_async = _auto(ctx, t, 1 | 4)
    assert p_sync == p_host == p_async, (p_sync, p_host, p_async)
    assert np.array_equal(o_sync, o_host) and np.array_equal(o_sync, o_async)
    assert len({bs for bs, _ in p_async}) >= 2, p_async                                  # more than one grid took part
    # the parameters again, against the oracle chain for one frame
    bs, cl = knee_mirror.select_parameters(orc.sweep(orc.gaussian3(frames[0])))
    assert p_async[0] == (bs, cl)
    # (the hook acts in the device choice; the synchronous forms re-select on the host for such frames, so the check here is
    # against the oracle: BS = last arg-max of the float16 entropies of CLAHE(blurred plane, 30, g), output = CLAHE(plane, 30, BS))
    monkeypatch.setenv("UWIP_ACLAHE_TEST_FORCE_CL", "30")
    ctx.prof_reset(); ctx.prof_enable(True)
    q_async, r_async = _auto(ctx, t, 1 | 4)
    prof = ctx.prof_results()
    ctx.prof_enable(False)
    monkeypatch.delenv("UWIP_ACLAHE_TEST_FORCE_CL")
    assert all(c == 30 for _, c in q_async), q_async
    # k_aclahe_exact_bs is ONE block per flagged frame walking the plane five times -- the worst case of the "no host
    # wait" stage, reached only by a degenerate fit (none in 1043 tables).  Its cost with EVERY frame flagged
    # is bounded here (one block per frame and grid, side by side; round 4: one per frame, 3.9 / 63 ms): < 4 ms at 270 x 480, < 30 ms at 1080p.
    ms, cnt = prof["k_aclahe_exact_bs"]
    print(f"k_aclahe_exact_bs, all {F} frames of {cols}x{rows} flagged: {ms / cnt:.2f} ms")
    assert ms / cnt < (4.0 if rows < 1000 else 30.0), ms / cnt
    for f in range(F if rows < 1000 else 2):
        filt = orc.gaussian3(frames[f])
        ent = np.array([orc.entropy(orc.clahe(filt, 30.0, g, g)) for g in (2, 4, 8, 16, 32)], np.float32).astype(np.float16)
        want = (2, 4, 8, 16, 32)[int(np.nonzero(ent == ent.max())[0][-1])]
        assert q_async[f][0] == want, (f, q_async[f], ent)
        assert np.array_equal(r_async[f], orc.clahe(frames[f], 30.0, want, want)), f
