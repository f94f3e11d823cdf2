"""GPU parity: histretch kernels (through the C ABI) vs the CPU oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch

from uwimageproc_amd import batch_of, preprocessing as pp, synth

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("shape", [(48, 64), (37, 53), (480, 640), (1, 1), (3, 5), (1080, 1920)])
def test_getHistogram_matches_oracle(ctx, orc, shape):
    img = synth.uw_frame(11, *shape)
    h = pp.getHistogram(ctx, _dev(img)).cpu().numpy()
    assert h.shape == (1, 3, 256)
    for c in range(3):
        assert np.array_equal(h[0, c], orc.getHistogram(img[..., c]))
    # single-channel plane
    v = np.ascontiguousarray(img[..., 1])
    h1 = pp.getHistogram(ctx, _dev(v)).cpu().numpy()
    assert np.array_equal(h1[0, 0], orc.getHistogram(v))


@pytest.mark.parametrize("kind", ["uw", "random", "two_level", "ramp", "constant"])
@pytest.mark.parametrize("letters", ["RGB", "B", "GRRB", "r", "RxGyB"])
def test_histretch_bit_exact(ctx, orc, kind, letters):
    img = synth.uw_frame(4, 96, 130) if kind == "uw" else synth.adversarial(kind, 96, 130)
    exp, rc = orc.histretch(img, letters)
    assert rc == 0
    t = _dev(img)
    pp.histretch(ctx, t, letters)
    assert np.array_equal(t.cpu().numpy(), exp)


def test_histretch_batch_and_config0_size(ctx, orc):
    # config[0] geometry (640x480) + a batch of distinct frames
    frames = synth.uw_batch(20, 5, 480, 640)
    t = _dev(frames)
    pp.histretch(ctx, t, "RGB")
    out = t.cpu().numpy()
    for f in range(frames.shape[0]):
        exp, _ = orc.histretch(frames[f], "RGB")
        assert np.array_equal(out[f], exp), f
    # thresholds at 640x480 are 6144 / 301056 (SURVEY A-2)
    assert np.float32(480 * 640 / 100.0) * 2 == 6144 and np.float32(480 * 640 / 100.0) * 98 == 301056


def test_histretch_strided_rows_and_unaligned(ctx, orc):
    # rows padded (step > cols*3) and a data pointer that is not 16-byte aligned
    rows, cols = 50, 67
    img = synth.uw_frame(8, rows, cols)
    exp, _ = orc.histretch(img, "RGB")
    step = cols * 3 + 13
    buf = torch.zeros(rows * step + 32, dtype=torch.uint8, device="cuda")
    for off in (0, 1, 16):
        buf.zero_()
        view = buf[off:off + rows * step].view(rows, step)[:, :cols * 3].view(rows, cols, 3)
        view.copy_(_dev(img))
        b = batch_of(view)
        assert b.step == step
        pp.histretch(ctx, view, "RGB")
        assert np.array_equal(view.cpu().numpy(), exp)
        # padding bytes untouched
        assert int(buf[off:off + rows * step].view(rows, step)[:, cols * 3:].max()) == 0


def test_imgChannelStretch_plane_and_lane(ctx, orc):
    img = synth.uw_frame(9, 64, 80)
    # split plane
    plane = np.ascontiguousarray(img[..., 2])
    exp = plane.copy()
    orc.imgChannelStretch(exp, 2, 98)
    t = _dev(plane)
    pp.imgChannelStretch(ctx, t, t, 2, 98)
    assert np.array_equal(t.cpu().numpy(), exp)
    # default percentiles 0/100 (preprocessing.h:66)
    exp = plane.copy()
    orc.imgChannelStretch(exp, 0, 100)
    t = _dev(plane)
    pp.imgChannelStretch(ctx, t)
    assert np.array_equal(t.cpu().numpy(), exp)
    # lane of a packed image
    exp3 = img.copy()
    orc.imgChannelStretch(exp3[..., 1], 5, 90)
    t3 = _dev(img)
    pp.imgChannelStretch(ctx, t3, None, 5, 90, channel=1)
    assert np.array_equal(t3.cpu().numpy(), exp3)


def test_stretch_lut_stage_tap(ctx, orc):
    img = synth.uw_frame(10, 120, 160)
    t = _dev(img)
    b = batch_of(t)
    hist = torch.zeros((1, 3, 256), dtype=torch.int32, device="cuda")
    lut = torch.zeros((3, 256), dtype=torch.uint8, device="cuda")
    bounds = torch.zeros((3, 2), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.call("uwip_getHistogram", C.byref(b), C.c_void_p(hist.data_ptr()))
    ctx.call("uwip_stretch_lut", C.c_void_p(hist.data_ptr()), 3, 120, 160, 2, 98,
             C.c_void_p(lut.data_ptr()), C.c_void_p(bounds.data_ptr()))
    ctx.sync()
    for c in range(3):
        el, lo, hi = orc.stretch_lut(orc.getHistogram(img[..., c]), 120, 160, 2, 98)
        assert np.array_equal(lut[c].cpu().numpy(), el)
        assert bounds[c].tolist() == [lo, hi]


@pytest.mark.parametrize("letters", ["V", "H", "Y", "RHG", "BCX", "SVR", "YH"])
@pytest.mark.parametrize("shape", [(48, 64), (37, 53)])
def test_colour_space_letters_as_written(ctx, orc, letters, shape):
    """HSV / YCrCb letters as the reference executes them: the 8-bit colour round trip of the image,
    interleaved in order with the BGR letters' stretches."""
    img = synth.uw_frame(11, *shape)
    t = _dev(img)
    pp.histretch(ctx, t, letters)
    exp, rc = orc.histretch(img, letters)
    assert rc == 0
    assert np.array_equal(t.cpu().numpy(), exp)


@pytest.mark.parametrize("space", [1, 2, 3, 4])
def test_cvtColor_both_directions_vs_oracle(ctx, orc, space):
    """cv::cvtColor BGR2{HSV,HLS,Lab,YCrCb} and back, 8UC3: bit-exact against the oracle's restatement (every 8-bit
    colour of a dense random sample, a real-looking frame, strided batches)."""
    rng = np.random.default_rng(space)
    for img in (rng.integers(0, 256, (96, 128, 3), dtype=np.uint8), synth.uw_frame(4, 135, 241)):
        fwd = pp.cvtColor(ctx, _dev(img), space).cpu().numpy()
        assert np.array_equal(fwd, orc.cvt_space(img, space)), space
        back = pp.cvtColor(ctx, _dev(fwd), space, to_bgr=True).cpu().numpy()
        assert np.array_equal(back, orc.cvt_space(fwd, space, True)), space
        if space == 3:
            # Lab -> BGR in both OpenCV forms: 3.4.x integer (the default, above) and 3.2 float; ALL 2^24 Lab triples would
            # take a while -- a dense random sample of them (most are out of gamut: the clamps are exercised too)
            lab = rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)
            for v32 in (False, True):
                got = pp.cvtColor(ctx, _dev(lab), 3, to_bgr=True, opencv32=v32).cpu().numpy()
                assert np.array_equal(got, orc.cvt_space(lab, 3, True, opencv32=v32)), v32
    batch = rng.integers(0, 256, (3, 40, 56, 3), dtype=np.uint8)
    got = pp.cvtColor(ctx, _dev(batch), space).cpu().numpy()
    for f in range(3):
        assert np.array_equal(got[f], orc.cvt_space(batch[f], space))


@pytest.mark.parametrize("letters", ["l", "h", "s", "L", "a", "b", "RlG", "LaV", "YhB"])
@pytest.mark.parametrize("fixed", [False, True])
def test_hls_lab_letters_and_fixed_order(ctx, orc, letters, fixed):
    """hsl / Lab letters: as written they leave the 8-bit colour round trip (B-3); with the fixed-order
    flag the stretch of the letter's plane is kept (convert, split / stretch / merge, convert back)."""
    img = synth.uw_frame(21, 72, 100)
    t = _dev(img)
    pp.histretch(ctx, t, letters, fixed_order=fixed)
    assert np.array_equal(t.cpu().numpy(), orc.histretch_ex(img, letters, fixed_order=fixed)), (letters, fixed)
    if any(c in "Lab" for c in letters):
        t = _dev(img)
        pp.histretch(ctx, t, letters, fixed_order=fixed, opencv32=True)             # UWIP_HISTRETCH_OPENCV32
        assert np.array_equal(t.cpu().numpy(), orc.histretch_ex(img, letters, fixed_order=fixed, opencv32=True)), (letters, fixed)


@pytest.mark.parametrize("letters", ["V", "HY", "RSC"])
def test_fixed_order_hsv_ycrcb(ctx, orc, letters):
    img = synth.uw_frame(22, 60, 84)
    t = _dev(img)
    pp.histretch(ctx, t, letters, fixed_order=True)
    assert np.array_equal(t.cpu().numpy(), orc.histretch_ex(img, letters, fixed_order=True))


def test_empty_and_errors(ctx):
    import uwimageproc_amd as uw
    e = torch.zeros((0, 8, 3), dtype=torch.uint8, device="cuda")
    pp.histretch(ctx, e, "RGB")            # empty input is a no-op
    img = _dev(synth.uw_frame(1, 16, 16))
    with pytest.raises(uw.UwipError):
        pp.imgChannelStretch(ctx, img, None, 2, 98, channel=5)


def test_full_size_properties_1080p_4k(ctx, orc):
    # size-independent properties at BASELINE sizes: histogram mass, idempotence of a
    # saturated stretch, and a sampled-frame oracle comparison
    for rows, cols in ((1080, 1920), (2160, 3840)):
        img = synth.uw_frame(30, rows, cols)
        t = _dev(img)
        h = pp.getHistogram(ctx, t)
        assert (h.sum(dim=2) == rows * cols).all()
        pp.histretch(ctx, t, "RGB")
        out = t.cpu().numpy()
        exp, _ = orc.histretch(img, "RGB")
        assert np.array_equal(out, exp)
        assert out.min() == 0 and out.max() == 255
