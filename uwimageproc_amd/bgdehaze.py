"""Host-side mirror of modules/bgdehaze/{BGDehaze,main}.py over the C ABI.

The reference functions take ``normI`` (float64 in [0,1], produced in
main.py:17 from the uint8 image).  The device path starts from the uint8 frame
itself (normI takes only 256 distinct values, so nothing is lost); every
function here therefore takes the uint8 BGR tensor ``I`` (H x W x 3 or
F x H x W x 3, CUDA) that ``cv2.imread`` would have produced.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._native import Context, batch_of


def _pre(t):
    torch.cuda.current_stream(t.device).synchronize()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def Background_light(ctx: Context, I: torch.Tensor, w: int = 15, return_index: bool = False):
    """BGDehaze.py:14-26 -> float64 [frames, 3] (BGR)."""
    b = batch_of(I)
    B = torch.empty((b.frames, 3), dtype=torch.float64, device=I.device)
    idx = torch.empty((b.frames, 2), dtype=torch.int32, device=I.device)
    _pre(I)
    ctx.call("uwip_dehaze_background_light", C.byref(b), int(w), _ptr(B), _ptr(idx))
    ctx.sync()
    return (B, idx) if return_index else B


def transmission_map(ctx: Context, I: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """BGDehaze.py:28-37 with B injected -> float64 [frames, 2, H, W] (blue, green)."""
    b = batch_of(I)
    B = B.to(torch.float64).reshape(b.frames, 3).contiguous()
    t = torch.empty((b.frames, 2, b.rows, b.cols), dtype=torch.float64, device=I.device)
    _pre(I)
    ctx.call("uwip_dehaze_transmission", C.byref(b), _ptr(B), _ptr(t))
    ctx.sync()
    return t


def guided_filter(ctx: Context, guide_u8: torch.Tensor, p: torch.Tensor, r: int = 40, eps: float = 1e-3) -> torch.Tensor:
    """guidedfilter.py:54-103 with guide = (guide_u8 - min)/(max - min)."""
    b = batch_of(guide_u8)
    p = p.to(torch.float64).reshape(b.frames, b.rows, b.cols).contiguous()
    q = torch.empty_like(p)
    _pre(guide_u8)
    ctx.call("uwip_guided_filter", C.byref(b), _ptr(p), int(r), C.c_double(eps), _ptr(q))
    ctx.sync()
    return q


DEHAZE_FULL, DEHAZE_GUARD_S = 1, 2


def dehaze(ctx: Context, I: torch.Tensor, w: int = 15, full: bool = True, B: torch.Tensor = None,
           want_refined_t: bool = False, want_float: bool = False, out: torch.Tensor = None,
           guard_s: bool = False):
    """main.py:14-20 (generate_results): uint8 BGR -> uint8 BGR.
    Returns ``out`` or a dict with the requested taps."""
    b = batch_of(I)
    if out is None:
        out = torch.empty_like(I)
    ob = batch_of(out)
    Bd = B.to(torch.float64).reshape(b.frames, 3).contiguous() if B is not None else None
    rt = torch.empty((b.frames, 2, b.rows, b.cols), dtype=torch.float64, device=I.device) if want_refined_t else None
    fo = torch.empty((b.frames, b.rows, b.cols, 3), dtype=torch.float64, device=I.device) if want_float else None
    _pre(I)
    flags = (DEHAZE_FULL if full else 0) | (DEHAZE_GUARD_S if guard_s else 0)
    ctx.call("uwip_dehaze", C.byref(b), C.byref(ob), int(w), flags, _ptr(Bd), _ptr(rt), _ptr(fo))
    ctx.sync()
    if want_refined_t or want_float:
        return {"out": out, "refined_t": rt, "float": fo}
    return out


def dehaze_histretch(ctx: Context, I: torch.Tensor, cChannel: str = "RGB", min_percent: int = 2, max_percent: int = 98,
                     w: int = 15, full: bool = True, guard_s: bool = False, fixed_order: bool = False,
                     out: torch.Tensor = None) -> torch.Tensor:
    """bgdehaze then histretch on its output (BASELINE configs[2]) as one chained call: the kernel that writes the
    dehazed bytes also histograms them.  Same bytes as ``dehaze`` followed by ``preprocessing.histretch``."""
    b = batch_of(I)
    if out is None:
        out = torch.empty_like(I)
    ob = batch_of(out)
    _pre(I)
    flags = (DEHAZE_FULL if full else 0) | (DEHAZE_GUARD_S if guard_s else 0)
    ctx.call("uwip_dehaze_histretch", C.byref(b), C.byref(ob), int(w), flags, cChannel.encode(), int(min_percent),
             int(max_percent), 1 if fixed_order else 0)
    ctx.sync()
    return out


def RC_correction(ctx: Context, I: torch.Tensor, w: int = 15, B: torch.Tensor = None) -> torch.Tensor:
    """BGDehaze.py:59-69 -> float64 [frames, H, W, 3]."""
    return dehaze(ctx, I, w, full=False, B=B, want_float=True)["float"]


def adaptiveExp_map(ctx: Context, I: torch.Tensor, w: int = 15, B: torch.Tensor = None) -> torch.Tensor:
    """BGDehaze.py:71-89 -> float64 [frames, H, W, 3]."""
    return dehaze(ctx, I, w, full=True, B=B, want_float=True)["float"]
