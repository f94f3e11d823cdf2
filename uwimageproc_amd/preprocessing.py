"""Host-side mirror of modules/common/preprocessing.{h,cpp} and the per-letter
loop of modules/histretch/src/histretch.cpp:217-254 over the C ABI.

Images are torch uint8 CUDA tensors laid out like cv::Mat (H x W x 3 BGR
interleaved, or H x W), optionally with a leading frame axis.  Like the
reference (cv::Mat headers passed by value share pixels), the stretch
functions mutate the caller's pixels in place.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._native import Context, batch_of, lib


def numChannel(c: str) -> int:
    """preprocessing.cpp:147-153."""
    return lib().uwip_numChannel(c.encode()[:1])


def numSpace(c: str) -> int:
    """preprocessing.cpp:155-161."""
    return lib().uwip_numSpace(c.encode()[:1])


def getHistogram(ctx: Context, img: torch.Tensor) -> torch.Tensor:
    """preprocessing.cpp:25-34 for every frame/channel: returns float32 counts
    [frames, channels, 256] (cv::calcHist's CV_32F histogram)."""
    b = batch_of(img)
    hist = torch.empty((b.frames, b.channels, 256), dtype=torch.int32, device=img.device)
    torch.cuda.current_stream(img.device).synchronize()
    ctx.call("uwip_getHistogram", C.byref(b), C.c_void_p(hist.data_ptr()))
    ctx.sync()
    return hist.to(torch.float32)


def imgChannelStretch(ctx: Context, imgOriginal: torch.Tensor, imgStretched: torch.Tensor = None,
                      lowerPercentile: int = 0, higherPercentile: int = 100, channel: int = 0) -> None:
    """preprocessing.cpp:74-105.  ``imgStretched`` must alias ``imgOriginal``
    (every reference call site passes the same Mat twice); ``channel`` selects
    the lane when the tensor is packed BGR instead of a split plane."""
    if imgStretched is not None and imgStretched.data_ptr() != imgOriginal.data_ptr():
        raise ValueError("imgStretched must share pixels with imgOriginal (in-place op)")
    b = batch_of(imgOriginal)
    torch.cuda.current_stream(imgOriginal.device).synchronize()
    ctx.call("uwip_imgChannelStretch", C.byref(b), int(channel), int(lowerPercentile), int(higherPercentile))
    ctx.sync()


def histretch(ctx: Context, src: torch.Tensor, cChannel: str, min_percent: int = 2, max_percent: int = 98,
              fixed_order: bool = False, opencv32: bool = False) -> None:
    """histretch.cpp:217-254 (CPU branch) on BGR frames, in place.  fixed_order=True keeps the stretch of the
    non-BGR letters (merge before converting back) instead of the reference's as-written round trip (B-3); opencv32=True
    converts Lab back to BGR as OpenCV 3.2 does (float form) instead of 3.4.x (integer form, the default)."""
    b = batch_of(src)
    torch.cuda.current_stream(src.device).synchronize()
    ctx.call("uwip_histretch_ex", C.byref(b), cChannel.encode(), int(min_percent), int(max_percent), (1 if fixed_order else 0) | (2 if opencv32 else 0))
    ctx.sync()


def cvtColor(ctx: Context, src: torch.Tensor, space: int, to_bgr: bool = False, opencv32: bool = False) -> torch.Tensor:
    """cv::cvtColor(src, COLOR_BGR2{HSV,HLS,Lab,YCrCb}) / the inverse, 8UC3; space = numSpace's index 1..4; opencv32: the
    Lab inverse of OpenCV 3.2 (float) instead of 3.4.x (integer)."""
    dst = torch.empty_like(src)
    sb, db = batch_of(src), batch_of(dst)
    torch.cuda.current_stream(src.device).synchronize()
    ctx.call("uwip_cvtColor_ex", C.byref(sb), C.byref(db), int(space), 1 if to_bgr else 0, 1 if opencv32 else 0)
    ctx.sync()
    return dst
