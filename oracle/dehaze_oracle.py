"""CPU restatement (numpy, float64) of the bgdehaze module.

TEST INFRASTRUCTURE ONLY (see oracle/uwip_oracle.c header): tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the
product path never does.

Follows modules/bgdehaze/BGDehaze.py and modules/bgdehaze/guidedfilter.py
(cited per function).  Pinned by tests/golden/dehaze_*.npz, which were
produced by importing the reference's own Python in the build container
(tools/make_goldens.py).  Two documented deviations from the reference's
letter, both from SURVEY.md Appendix B:
  B-9  background-light ties: first (row-major) index instead of the
       reference's unstable argsort order;
  B-10 the background light is computed once, not twice.
The adaptiveExp_map tail needs cv2.cvtColor(BGR2YCrCb), which lives in OpenCV
(absent): its 8-bit fixed-point formula is restated here -> that stage is
"parity unpinned".
"""
import numpy as np


def normalize_input(I_u8):
    """modules/bgdehaze/main.py:16-17: uint8 BGR -> float64, GLOBAL min/max."""
    I = np.asarray(I_u8)
    mn, mx = I.min(), I.max()
    return (I - mn) / (mx - mn)          # uint8 - uint8 -> uint8, then true divide -> float64


def _window_reduce(a, w, fn):
    """fn over the zero-padded w x w window anchored as BGDehaze.py:16-21 does
    (pad floor(w/2) on every side, window = padded[y:y+w, x:x+w])."""
    pad = w // 2
    p = np.pad(a, ((pad, pad), (pad, pad)), "constant")
    M, N = a.shape
    # separable: rows then columns (max/min are exact, so order is immaterial)
    tmp = p[:, 0:N].copy()
    for k in range(1, w):
        tmp = fn(tmp, p[:, k:k + N])
    out = tmp[0:M].copy()
    for k in range(1, w):
        out = fn(out, tmp[k:k + M])
    return out


def background_light(normI, w=15):
    """BGDehaze.py:14-26 (ties: first index)."""
    mxB = _window_reduce(normI[:, :, 0], w, np.maximum)
    mxG = _window_reduce(normI[:, :, 1], w, np.maximum)
    mxR = _window_reduce(normI[:, :, 2], w, np.maximum)
    D0 = (mxR - mxB).ravel()
    D1 = (mxR - mxG).ravel()
    i0, i1 = int(np.argmin(D0)), int(np.argmin(D1))
    flatI = normI.reshape(-1, 3)
    return (flatI[i0] + flatI[i1]) / 2.0, (i0, i1)


def transmission_map(normI, B, w=15):
    """BGDehaze.py:28-37 with B injected."""
    q = normI / B
    t0 = 1 - _window_reduce(q[:, :, 0], w, np.minimum)
    t1 = 1 - _window_reduce(q[:, :, 1], w, np.minimum)
    return np.stack([t0, t1], axis=2)


def boxfilter(I, r):
    """guidedfilter.py:23-51: sums over the in-image part of a (2r+1)^2 window
    (same cumsum differences, hence the same rounding)."""
    M, N = I.shape
    assert M >= 2 * r + 1 and N >= 2 * r + 1, "guided filter needs both dims >= 2r+1"
    S = np.zeros((M + 1, N))
    np.cumsum(I, axis=0, out=S[1:])
    hi = np.minimum(np.arange(M) + r, M - 1) + 1
    lo = np.maximum(np.arange(M) - r, 0)
    d = S[hi] - S[lo]
    S2 = np.zeros((M, N + 1))
    np.cumsum(d, axis=1, out=S2[:, 1:])
    hi = np.minimum(np.arange(N) + r, N - 1) + 1
    lo = np.maximum(np.arange(N) - r, 0)
    return S2[:, hi] - S2[:, lo]


def guided_filter(I, p, r=40, eps=1e-3):
    """guidedfilter.py:54-103 (colour guided filter, He et al. ECCV10)."""
    M, N = p.shape
    base = boxfilter(np.ones((M, N)), r)
    means = [boxfilter(I[:, :, i], r) / base for i in range(3)]
    mean_p = boxfilter(p, r) / base
    means_IP = [boxfilter(I[:, :, i] * p, r) / base for i in range(3)]
    covIP = [means_IP[i] - means[i] * mean_p for i in range(3)]
    var = {}
    for i in range(3):
        for j in range(i, 3):
            var[(i, j)] = boxfilter(I[:, :, i] * I[:, :, j], r) / base - means[i] * means[j]
    Sigma = np.empty((M, N, 3, 3))
    for i in range(3):
        for j in range(3):
            Sigma[:, :, i, j] = var[(min(i, j), max(i, j))]
    Sigma = Sigma + eps * np.eye(3)
    cov = np.stack(covIP, axis=2)
    a = np.einsum("mni,mnij->mnj", cov, np.linalg.inv(Sigma))       # eq 14
    b = mean_p - a[:, :, 0] * means[0] - a[:, :, 1] * means[1] - a[:, :, 2] * means[2]   # eq 15
    q = (boxfilter(a[:, :, 0], r) * I[:, :, 0] + boxfilter(a[:, :, 1], r) * I[:, :, 1]
         + boxfilter(a[:, :, 2], r) * I[:, :, 2] + boxfilter(b, r)) / base           # eq 16
    return q


def refined_t(normI, B, tmin=0.2, r=40, eps=1e-3):
    """BGDehaze.py:39-48 (always w=15: B-10)."""
    t = transmission_map(normI, B, 15)
    tb = guided_filter(normI, np.maximum(t[:, :, 0], tmin), r, eps)
    tg = guided_filter(normI, np.maximum(t[:, :, 1], tmin), r, eps)
    return tb, tg


def _minmax(a):
    return (a - a.min()) / (a.max() - a.min())


def dehazed_BG(normI, B):
    """BGDehaze.py:50-57."""
    tb, tg = refined_t(normI, B)
    Jb = (normI[:, :, 0] - B[0]) / tb + B[0]
    Jg = (normI[:, :, 1] - B[1]) / tg + B[1]
    return _minmax(Jb), _minmax(Jg)


def RC_correction(normI, w=15, B=None):
    """BGDehaze.py:59-69.  B may be injected (parity tests, B-9)."""
    if B is None:
        B, _ = background_light(normI, w)
    nJb, nJg = dehazed_BG(normI, B)
    avgRr = 1.5 - np.average(nJb.ravel()) - np.average(nJg.ravel())
    compCoeff = avgRr / np.average(normI[:, :, 2].ravel())
    Rrec = normI[:, :, 2] * compCoeff
    restored = np.zeros(normI.shape)
    restored[:, :, 0] = nJb
    restored[:, :, 1] = nJg
    restored[:, :, 2] = _minmax(Rrec)
    return restored


def bgr2ycrcb_u8(img):
    """cv2.cvtColor(COLOR_BGR2YCrCb) for 8-bit input, OpenCV 3.x fixed point
    (yuv_shift = 14).  parity unpinned (OpenCV-internal)."""
    b = img[:, :, 0].astype(np.int64)
    g = img[:, :, 1].astype(np.int64)
    r = img[:, :, 2].astype(np.int64)
    Y = (b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14
    Cr = ((r - Y) * 11682 + (128 << 14) + (1 << 13)) >> 14
    Cb = ((b - Y) * 9241 + (128 << 14) + (1 << 13)) >> 14
    out = np.stack([Y, Cr, Cb], axis=2)
    return np.clip(out, 0, 255).astype(np.uint8)


def adaptiveExp_tail(normI, restored, r=40, eps=1e-3, guard_s=False):
    """BGDehaze.py:75-89, given RC_correction's output.  The uint8 casts at
    :75-76 TRUNCATE, so this stage is ill-conditioned wherever restored*255
    sits on an integer (the linearly mapped red channel does so routinely):
    tests feed it the device's own `restored` to check the tail in isolation.
    guard_s mirrors UWIP_DEHAZE_GUARD_S (S = 1 where it would be 0/0)."""
    R = (restored * 255).astype(np.uint8)
    I = (normI * 255).astype(np.uint8)
    YjCrCb = bgr2ycrcb_u8(R)
    YiCrCb = bgr2ycrcb_u8(I)
    normYj = (YjCrCb - YjCrCb.min()) / (YjCrCb.max() - YjCrCb.min())
    normYi = (YiCrCb - YiCrCb.min()) / (YiCrCb.max() - YiCrCb.min())
    Yi = normYi[:, :, 0]
    Yj = normYj[:, :, 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        num, den = Yj * Yi + 0.3 * Yi ** 2, Yj ** 2 + 0.3 * Yi ** 2
        S = num / den
        if guard_s:
            S = np.where(den == 0, 1.0, S)
        refinedS = guided_filter(normYi, S, r, eps)
        out = restored * refinedS[:, :, None]
        return (out - out.min()) / (out.max() - out.min())


def adaptiveExp_map(normI, w=15, B=None, r=40, eps=1e-3, guard_s=False):
    """BGDehaze.py:71-89."""
    return adaptiveExp_tail(normI, RC_correction(normI, w, B), r, eps, guard_s)


def to_u8(restored):
    """cv2.imwrite(dest, restored*255) (main.py:19): convertTo(CV_8U) = RNE + saturate."""
    v = np.nan_to_num(restored * 255, nan=-1.0)          # cvRound(NaN) = INT_MIN -> saturates to 0
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def bgdehaze_u8(I_u8, w=15, full=True):
    """main.py:14-20 end to end: uint8 BGR -> uint8 BGR."""
    normI = normalize_input(I_u8)
    out = adaptiveExp_map(normI, w) if full else RC_correction(normI, w)
    return to_u8(out)
