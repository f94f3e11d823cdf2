"""Host-side mirror of the aclahe module over the C ABI:
cv::CLAHE as driven by modules/aclahe/src/aclahe.cpp:175-187, aclaheEntropy
(:228-248) and the 5 x 51 sweep (:160-193).  Planes are torch uint8 CUDA
tensors [H, W] or [F, H, W]."""
from __future__ import annotations

import ctypes as C

import torch

from ._native import Context, batch_of

BLOCK_SIZES = (2, 4, 8, 16, 32)                       # aclahe.cpp:161
CLIP_LIMITS = tuple(0.5 * i for i in range(51))       # aclahe.cpp:163-165,181


def _pre(t):
    torch.cuda.current_stream(t.device).synchronize()


def bgr_to_v(ctx: Context, bgr: torch.Tensor) -> torch.Tensor:
    """cvtColor(BGR2HSV) + split -> channels[2] (aclahe.cpp:152-154)."""
    b = batch_of(bgr)
    shape = (b.frames, b.rows, b.cols) if bgr.dim() == 4 else (b.rows, b.cols)
    v = torch.empty(shape, dtype=torch.uint8, device=bgr.device)
    vb = batch_of(v)
    _pre(bgr)
    ctx.call("uwip_bgr_to_v", C.byref(b), C.byref(vb))
    ctx.sync()
    return v


class CLAHE:
    """cv::createCLAHE() look-alike: setClipLimit / setTilesGridSize / apply."""

    def __init__(self, ctx: Context, clipLimit: float = 40.0, tileGridSize=(8, 8), residual_rule: int = 0):
        self.ctx, self.clipLimit, self.tileGridSize, self.residual_rule = ctx, float(clipLimit), tuple(tileGridSize), residual_rule

    def setClipLimit(self, cl: float):
        self.clipLimit = float(cl)

    def setTilesGridSize(self, size):
        self.tileGridSize = tuple(size)

    def apply(self, src: torch.Tensor, dst: torch.Tensor = None) -> torch.Tensor:
        if dst is None:
            dst = torch.empty_like(src)
        sb, db = batch_of(src), batch_of(dst)
        _pre(src)
        self.ctx.call("uwip_clahe", C.byref(sb), C.byref(db), C.c_double(self.clipLimit),
                      int(self.tileGridSize[0]), int(self.tileGridSize[1]), int(self.residual_rule))
        self.ctx.sync()
        return dst

    def luts(self, src: torch.Tensor) -> torch.Tensor:
        sb = batch_of(src)
        gx, gy = self.tileGridSize
        out = torch.empty((sb.frames, gy * gx, 256), dtype=torch.uint8, device=src.device)
        _pre(src)
        self.ctx.call("uwip_clahe_luts", C.byref(sb), C.c_double(self.clipLimit), int(gx), int(gy),
                      int(self.residual_rule), C.c_void_p(out.data_ptr()))
        self.ctx.sync()
        return out


def clahe_per_frame(ctx: Context, src: torch.Tensor, clipLimits, grids, dst: torch.Tensor = None,
                    residual_rule: int = 0) -> torch.Tensor:
    """Final `createCLAHE(CL,(BS,BS)).apply` with per-frame (BS, CL)."""
    if dst is None:
        dst = torch.empty_like(src)
    sb, db = batch_of(src), batch_of(dst)
    n = sb.frames
    cl = (C.c_double * n)(*[float(x) for x in clipLimits])
    gr = (C.c_int32 * n)(*[int(x) for x in grids])
    _pre(src)
    ctx.call("uwip_clahe_per_frame", C.byref(sb), C.byref(db), cl, gr, int(residual_rule))
    ctx.sync()
    return dst


def GaussianBlur3(ctx: Context, img: torch.Tensor, rounding_rule: int = 0) -> torch.Tensor:
    """cv2.GaussianBlur(img, (3,3), 0) on 8-bit planes (ACLAHE.py:15)."""
    dst = torch.empty_like(img)
    sb, db = batch_of(img), batch_of(dst)
    _pre(img)
    ctx.call("uwip_GaussianBlur3", C.byref(sb), C.byref(db), int(rounding_rule))
    ctx.sync()
    return dst


ACLAHE_PREFILTER = 1


def auto(ctx: Context, src: torch.Tensor, dst: torch.Tensor = None, residual_rule: int = 0, prefilter: bool = True):
    """The whole aclahe stage on 8-bit planes: parameter search + final CLAHE.  prefilter=True is ParametrosACLAHE
    (ACLAHE.py:9-129: search on the 3x3-blurred plane, final apply on the plane itself, python/main.py:19-20);
    prefilter=False is the C++ driver's form (aclahe.cpp:152-187).  Returns (dst, [(BS, CL), ...])."""
    if dst is None:
        dst = torch.empty_like(src)
    sb, db = batch_of(src), batch_of(dst)
    n = sb.frames
    bs, cl = (C.c_int32 * n)(), (C.c_int32 * n)()
    _pre(src)
    ctx.call("uwip_aclahe_auto_ex", C.byref(sb), C.byref(db), int(residual_rule), ACLAHE_PREFILTER if prefilter else 0, bs, cl)
    ctx.sync()
    return dst, list(zip(bs, cl))


def aclaheEntropy(ctx: Context, img: torch.Tensor) -> torch.Tensor:
    """aclahe.cpp:228-248 per frame -> float32 [frames]."""
    b = batch_of(img)
    out = torch.empty((b.frames,), dtype=torch.float32, device=img.device)
    _pre(img)
    ctx.call("uwip_entropy", C.byref(b), C.c_void_p(out.data_ptr()))
    ctx.sync()
    return out


def sweep(ctx: Context, plane: torch.Tensor, residual_rule: int = 0) -> torch.Tensor:
    """aclahe.cpp:160-193: entropy table [frames, 5, 51] (grid x clip limit)."""
    b = batch_of(plane)
    out = torch.empty((b.frames, 5, 51), dtype=torch.float32, device=plane.device)
    _pre(plane)
    ctx.call("uwip_aclahe_sweep", C.byref(b), int(residual_rule), C.c_void_p(out.data_ptr()))
    ctx.sync()
    return out


def sweep_histograms(ctx: Context, plane: torch.Tensor, residual_rule: int = 0):
    """The sweep with its tap: (entropy [frames, 5, 51], histograms [frames, 5, 51, 256] of every CLAHE output)."""
    b = batch_of(plane)
    out = torch.empty((b.frames, 5, 51), dtype=torch.float32, device=plane.device)
    hist = torch.empty((b.frames, 5, 51, 256), dtype=torch.int32, device=plane.device)
    _pre(plane)
    ctx.call("uwip_aclahe_sweep_hist", C.byref(b), int(residual_rule), C.c_void_p(out.data_ptr()), C.c_void_p(hist.data_ptr()))
    ctx.sync()
    return out, hist


# ---------------------------------------------------------------------------
# C4: parameter selection, modules/aclahe/python/ACLAHE.py:66-129 +
# functions.py:49-93.  The reference does this on the host with scipy; so does
# this mirror (same calls: curve_fit p0=(7,0.4,0.9,5), splrep/splev).  The
# sweep loop of ACLAHE.py:40-47 is broken by indentation; the
# evident intent is implemented: entropy-vs-CL curves sampled at CL = 0.5 ...
# 24.5 (49 samples, functions.graficar slices [2:51]).
# ---------------------------------------------------------------------------
def _dexp(x, p0, p1, p2, p3):
    import numpy as np
    return p0 * np.exp(-p1 * x) + p2 * np.exp(-p3 * x)


def knee_index(xs, ys) -> int:
    """DerivadaY + DerivadaX + Curvatura (functions.py:49-93) -> arg-max index, or -1
    when curve_fit does not converge (the reference would raise there)."""
    import warnings

    import numpy as np
    from scipy.interpolate import splev, splrep
    from scipy.optimize import curve_fit

    u = np.linspace(1, 49, 49)
    p0 = (7, 0.4, 0.9, 5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            popt, _ = curve_fit(_dexp, u, ys, p0)
            x22 = np.linspace(1, 25, 25)
            tck = splrep(x22, _dexp(x22, *popt))
            x222 = np.linspace(1, 25, 49)
            y220 = splev(x222, tck, der=1)
            y221 = splev(x222, tck, der=2)
            popt2, _ = curve_fit(_dexp, u, xs, p0)
            tck2 = splrep(x22, _dexp(x22, *popt2))
            y223 = splev(x222, tck2, der=1)
            y225 = splev(x222, tck2, der=2)
        except Exception:
            return -1
        k = np.abs(y223 * y221 - y220 * y225) / np.power(np.power(y223, 2) + np.power(y220, 2), 1.5)
        return int(np.argmax(k))


def select_parameters(table, entropy_at=None):
    """(BS, CL) from one frame's 5 x 51 entropy table (ACLAHE.py:66-129).
    CL is the largest of the five curvature arg-max INDICES (:92-96, as written);
    BS is the block size whose entropy at clip limit CL is largest, compared in
    float16 with the LAST maximum winning (:102-125).  ``entropy_at(bs_index, cl)``
    supplies entropies for clip limits outside the swept grid."""
    import numpy as np

    table = np.asarray(table, dtype=np.float32)
    xs = (np.arange(51, dtype=np.float32) * 0.5)[1:50]
    idx = [knee_index(xs, table[g][1:50]) for g in range(5)]
    d = max(idx)
    if d < 0:
        d = 0
    ent = np.zeros(5, np.float16)
    for g in range(5):
        if 2 * d <= 50:
            ent[g] = table[g][2 * d]
        else:
            ent[g] = entropy_at(g, float(d)) if entropy_at is not None else table[g][50]
    w = int(np.nonzero(ent == ent.max())[0][-1])
    return BLOCK_SIZES[w], int(d)


def ParametrosACLAHE(ctx: Context, imagen: torch.Tensor, residual_rule: int = 0, prefilter: bool = True):
    """ACLAHE.py:9-129 for one 8-bit plane (or a batch): returns a list of (BS, CL).  The search runs on
    imgfilt = GaussianBlur(imagen, (3,3), 0) (:15) unless prefilter=False (the C++ driver's unfiltered sweep)."""
    if prefilter:
        imagen = GaussianBlur3(ctx, imagen, residual_rule)
    tab = sweep(ctx, imagen, residual_rule).cpu().numpy()
    planes = imagen if imagen.dim() == 3 else imagen[None]
    out = []
    for f in range(tab.shape[0]):
        def entropy_at(g, cl, f=f):
            c = CLAHE(ctx, cl, (BLOCK_SIZES[g],) * 2, residual_rule)
            return float(aclaheEntropy(ctx, c.apply(planes[f])).cpu()[0])
        out.append(select_parameters(tab[f], entropy_at))
    return out
