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


def ParametrosACLAHE(ctx: Context, imagen: torch.Tensor, residual_rule: int = 0, prefilter: bool = True):
    """ACLAHE.py:9-129 for one 8-bit plane (or a batch): returns a list of (BS, CL).  The search runs on
    imgfilt = GaussianBlur(imagen, (3,3), 0) (:15) unless prefilter=False (the C++ driver's unfiltered sweep).
    The choice is the library's (uwip_aclahe_auto_ex: device sweep + the native restatement of functions.py:49-93,
    entropies at clip limits outside the swept grid evaluated exactly); there is no scipy on the product path."""
    _, params = auto(ctx, imagen, residual_rule=residual_rule, prefilter=prefilter)
    return params
