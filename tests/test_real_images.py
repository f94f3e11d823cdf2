"""The reference ships three 1920x1080 underwater photographs (modules/bgdehaze/img/) and ONE output image
(modules/bgdehaze/result/BUL_T1A_0209.jpg = main.py with the default w = 15 on the second of them, saved as JPEG).
Two more outputs are there under other names: result/restoredFiltered.png and restoredFiltered2.png are main.py's
output, as lossless PNG, for the first and the third photograph (identified by content).  These are the only real
data and the only end-to-end input/output pairs the reference holds for this path (tests/golden/real/, copied /
cropped by tools/make_real_fixtures.py).

  * CPU: the oracle's whole chain (including the restated cv2 BGR2YCrCb, otherwise "parity unpinned") against the
    reference's own result -- a loose pin, bounded by JPEG coding of input and output and by the tie order of the
    background light.
  * GPU: uwip_dehaze against the oracle on all three photographs at full size, float stages 1e-9."""
import json
import os

import numpy as np
import pytest

import _knee_mirror as knee_mirror   # the scipy mirror of functions.py:49-93 (a checker: lives with the tests)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REAL = os.path.join(ROOT, "tests", "golden", "real")
INPUTS = ["in_BUL_T1A_0028.jpg", "in_BUL_T1A_0209.jpg", "in_PIS_T1A_259.jpg"]


def load_bgr(name):
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(os.path.join(REAL, name)).convert("RGB"))[:, :, ::-1])


def closeness(a, b):
    d = np.abs(a.astype(int) - b.astype(int))
    corr = [float(np.corrcoef(a[:, :, c].ravel(), b[:, :, c].ravel())[0, 1]) for c in range(3)]
    return float(d.mean()), [float(d[:, :, c].mean()) for c in range(3)], corr


def test_manifest_matches_files():
    import hashlib
    man = json.load(open(os.path.join(REAL, "MANIFEST.json")))
    for name, e in man.items():
        assert hashlib.sha256(open(os.path.join(REAL, name), "rb").read()).hexdigest() == e["sha256"]


def test_oracle_reproduces_the_references_own_result(orc):
    """adaptiveExp_map as written (no S guard), w = 15.
    With the product's first-index tie rule for the background light the result is the same picture with a slightly
    different cast (mean |diff| < 7 levels, correlation > 0.99 per channel); with the background light of the two tied
    pixels numpy's own argsort picks (BGDehaze.py:24; recorded in b9_argsort.json) it matches to JPEG noise
    (mean |diff| < 2.5 levels per channel)."""
    img, ref = load_bgr("in_BUL_T1A_0209.jpg"), load_bgr("ref_result_BUL_T1A_0209.jpg")
    b9 = json.load(open(os.path.join(REAL, "b9_argsort.json")))
    out, tap = orc.dehaze(img, 15, full=True, guard_s=False, taps=("idx",))
    mean, per, corr = closeness(out, ref)
    assert mean < 7.0 and min(corr) > 0.99, (mean, per, corr)
    assert b9["tie_counts"][0] > 1 and b9["tie_counts"][1] > 1        # the minima ARE tied on this photograph
    out2, _ = orc.dehaze(img, 15, full=True, guard_s=False, B=np.array(b9["B_argsort"]))
    mean2, per2, corr2 = closeness(out2, ref)
    assert max(per2) < 2.5 and min(corr2) > 0.99, (mean2, per2, corr2)


def load_crop(name):
    from PIL import Image
    c = json.load(open(os.path.join(REAL, "crops.json")))
    ref = np.ascontiguousarray(np.asarray(Image.open(os.path.join(REAL, name)).convert("RGB"))[:, :, ::-1])
    (y0, y1), (x0, x1) = c["window_rows"], c["window_cols"]
    return ref, (slice(y0, y1), slice(x0, x1)), c["files"][name]


LOSSLESS = [("ref_result_BUL_T1A_0028_crop.png", 1.1, 0.80), ("ref_result_PIS_T1A_259_crop.png", 2.6, 0.45)]


def _lossless_stats(out, ref):
    d = np.abs(out.astype(int) - ref.astype(int))
    return float(d.mean()), float((d <= 1).mean()), float((d == 0).mean())


@pytest.mark.parametrize("name,mean_bound,within1_bound", LOSSLESS)
def test_oracle_vs_the_references_lossless_outputs(orc, name, mean_bound, within1_bound):
    """result/restoredFiltered.png and restoredFiltered2.png are main.py's output (PNG, lossless) for BUL_T1A_0028 and
    PIS_T1A_259 (identified by content).  With only the JPEG decoder of the INPUT and the tie order of the background
    light between the oracle and the reference's run, the whole chain agrees to mean |diff| 0.8 (2.1) levels, 86 % (54 %)
    of the window's bytes within one level, 54 % (33 %) identical."""
    ref, win, src = load_crop(name)
    out, _ = orc.dehaze(load_bgr(src), 15, full=True, guard_s=False)
    mean, within1, exact = _lossless_stats(out[win], ref)
    assert mean < mean_bound and within1 > within1_bound, (name, mean, within1, exact)


@pytest.mark.gpu
@pytest.mark.parametrize("name,mean_bound,within1_bound", LOSSLESS)
def test_device_vs_the_references_lossless_outputs(ctx, name, mean_bound, within1_bound):
    import torch
    from uwimageproc_amd import bgdehaze as bg
    ref, win, src = load_crop(name)
    out = bg.dehaze(ctx, torch.from_numpy(load_bgr(src)).cuda(), 15, full=True).cpu().numpy()
    mean, within1, exact = _lossless_stats(out[win], ref)
    assert mean < mean_bound and within1 > within1_bound, (name, mean, within1, exact)


@pytest.mark.gpu
@pytest.mark.parametrize("name", INPUTS)
def test_device_vs_oracle_on_the_references_photographs(ctx, orc, name):
    import _dehaze_check
    img = load_bgr(name)
    rep, _ = _dehaze_check.check_frame(ctx, orc, img, guard=False, what=name)
    print(name, rep)


@pytest.mark.gpu
def test_device_reproduces_the_references_own_result(ctx, orc):
    """uwip_dehaze(FULL, as written) against the image the reference's authors saved: the loose end-to-end pin, same
    bounds as the oracle holds (test_oracle_reproduces_the_references_own_result)."""
    import torch
    from uwimageproc_amd import bgdehaze as bg
    img, ref = load_bgr("in_BUL_T1A_0209.jpg"), load_bgr("ref_result_BUL_T1A_0209.jpg")
    b9 = json.load(open(os.path.join(REAL, "b9_argsort.json")))
    t = torch.from_numpy(img).cuda()
    out = bg.dehaze(ctx, t, 15, full=True).cpu().numpy()
    mean, per, corr = closeness(out, ref)
    assert mean < 7.0 and min(corr) > 0.99, (mean, per, corr)
    out2 = bg.dehaze(ctx, t, 15, full=True, B=torch.tensor(b9["B_argsort"], dtype=torch.float64).cuda()).cpu().numpy()
    mean2, per2, corr2 = closeness(out2, ref)
    assert max(per2) < 2.5 and min(corr2) > 0.99, (mean2, per2, corr2)


@pytest.mark.gpu
def test_aclahe_on_the_references_crowd_image(ctx, orc):
    """modules/aclahe/python/main.py:  img = imread('crowd.png', 0);  BS, CL = ParametrosACLAHE(img);
    createCLAHE(CL, (BS, BS)).apply(img)  -- the reference's own input (its output, clahe_2.jpg, is not shipped):
    device against the oracle driven the same way, both forms of the stage."""
    import torch
    from PIL import Image
    from uwimageproc_amd import aclahe
    img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(REAL, "in_aclahe_crowd.png")).convert("L")))
    assert img.shape == (600, 800)
    t = torch.from_numpy(img).cuda()
    for prefilter in (True, False):
        dst, params = aclahe.auto(ctx, t, prefilter=prefilter)
        src = orc.gaussian3(img) if prefilter else img
        tab = orc.sweep(src)
        got_tab = aclahe.sweep(ctx, torch.from_numpy(src).cuda()).cpu().numpy()[0]
        assert np.abs(got_tab - tab).max() <= 1e-5
        bs, cl = knee_mirror.select_parameters(tab)
        assert params[0] == (bs, cl), (prefilter, params, (bs, cl))
        assert np.array_equal(dst.cpu().numpy(), orc.clahe(img, float(cl), bs, bs))
