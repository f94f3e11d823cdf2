"""Host-side mirror of modules/videostrip/{include/videostrip.hpp,src/videostrip.cpp}
and of the selector loop in src/main.cpp:300-394, over the C ABI.

``keyframe`` mirrors ``struct keyframe`` (videostrip.hpp:62-68): the key frame image
plus its cached keypoints/descriptors (here: a slot of a device feature set).
``videoWidth`` / ``videoHeight`` are module globals as in the reference
(main.cpp:45-49, extern in videostrip.cpp:29-33)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._native import Context, batch_of

TARGET_WIDTH, TARGET_HEIGHT = 640, 480        # videostrip.hpp:48-49
OVERLAP_MIN = 0.4                             # videostrip.hpp:50
DEFAULT_KWINDOW = 11                          # videostrip.hpp:51

videoWidth, videoHeight = 0, 0                # main.cpp:45-46

KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("response", "f4"), ("level", "i4"), ("xi", "i4"), ("yi", "i4"),
                     ("co", "f4"), ("si", "f4")])


class Features:
    """Opaque device feature set (uwip_features): `capacity` frame slots."""

    def __init__(self, ctx: Context, capacity: int):
        self.ctx, self.capacity = ctx, capacity
        h = C.c_void_p()
        ctx.call("uwip_features_create", int(capacity), C.byref(h))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._l.uwip_features_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def detect(self, frames: torch.Tensor, first_slot: int = 0, upright: bool = False, relative_threshold: bool = False):
        """upright: SURF's `upright` parameter (no orientation estimate); the reference runs SURF oriented.
        relative_threshold: detector threshold relative to the frame's contrast (UWIP_OVERLAP_RELATIVE_THRESHOLD) instead of
        the fixed 1e-3."""
        b = batch_of(frames)
        torch.cuda.current_stream(frames.device).synchronize()
        self.ctx.call("uwip_overlap_detect_ex", C.byref(b), self._h, int(first_slot), (1 if upright else 0) | (16 if relative_threshold else 0))

    def download(self, slot: int):
        kps = np.zeros(2048, KP_DTYPE)
        desc = np.zeros((2048, 64), np.uint8)
        n = C.c_int32(0)
        self.ctx.call("uwip_features_download", self._h, int(slot), C.c_void_p(kps.ctypes.data), C.c_void_p(desc.ctypes.data), C.byref(n))
        return kps[:n.value].copy(), desc[:n.value].copy()


OVERLAP_MIN6 = 8          # uwip.h UWIP_OVERLAP_MIN6: >= 6 RANSAC inliers instead of the reference's ">= 4 good matches" rule (the default)


def match_pairs(ctx: Context, fq: Features, ft: Features, pair_q, pair_t, vw: int, vh: int, seed: int = 1,
                want_matches: bool = False, min6: bool = False):
    """kNN(2) + ratio test + homography + overlapArea for (object slot, key slot) pairs."""
    n = len(pair_q)
    dev = torch.device("cuda", ctx.device)
    ratio = torch.empty((n,), dtype=torch.float32, device=dev)
    info = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    H = torch.zeros((n, 9), dtype=torch.float64, device=dev)
    midx = torch.full((n, 2048, 2), -1, dtype=torch.int32, device=dev) if want_matches else None
    mdist = torch.full((n, 2048, 2), -1, dtype=torch.int32, device=dev) if want_matches else None
    pq = (C.c_int32 * n)(*[int(v) for v in pair_q])
    pt = (C.c_int32 * n)(*[int(v) for v in pair_t])
    torch.cuda.synchronize()
    ctx.call("uwip_overlap_match_ex", fq._h, ft._h, pq, pt, n, int(vw), int(vh), int(seed), OVERLAP_MIN6 if min6 else 0,
             C.c_void_p(ratio.data_ptr()),
             C.c_void_p(info.data_ptr()), C.c_void_p(H.data_ptr()),
             C.c_void_p(midx.data_ptr()) if want_matches else None, C.c_void_p(mdist.data_ptr()) if want_matches else None)
    ctx.sync()
    out = {"ratio": ratio, "info": info, "H": H.reshape(n, 3, 3)}
    if want_matches:
        out["idx"], out["dist"] = midx, mdist
    return out


class keyframe:
    """struct keyframe (videostrip.hpp:62-68)."""

    def __init__(self, ctx: Context, img: torch.Tensor):
        self.ctx = ctx
        self.img = img
        self.new_img = True
        self.feats = Features(ctx, 1)


def calcOverlap(ctx: Context, kframe: keyframe, img_object: torch.Tensor, seed: int = 1) -> float:
    """videostrip.cpp:192-289.  ``img_object`` / ``kframe.img`` are full-resolution BGR frames
    (the 640-wide resize of main.cpp:311 happens inside).  Returns the overlap ratio, -1 for
    empty input, -2.0 when no homography can be estimated."""
    if img_object is None or kframe.img is None or img_object.numel() == 0 or kframe.img.numel() == 0:
        print(" --(!) Error reading images ")
        return -1.0
    if kframe.new_img:
        kframe.feats.detect(kframe.img, 0)
        kframe.new_img = False
    obj = Features(ctx, 1)
    obj.detect(img_object, 0)
    r = match_pairs(ctx, obj, kframe.feats, [0], [0], videoWidth, videoHeight, seed)
    obj.close()
    return float(r["ratio"].cpu()[0])


def overlapArea(ctx: Context, H) -> float:
    """videostrip.cpp:291-319 for one 3x3 homography (array-like, double)."""
    Hd = torch.as_tensor(np.asarray(H, dtype=np.float64).reshape(1, 9)).cuda()
    out = torch.empty((1,), dtype=torch.float32, device=Hd.device)
    torch.cuda.synchronize()
    ctx.call("uwip_overlapArea", C.c_void_p(Hd.data_ptr()), 1, int(videoWidth), int(videoHeight), C.c_void_p(out.data_ptr()), None)
    ctx.sync()
    return float(out.cpu()[0])


def resize_bgr(ctx: Context, frame: torch.Tensor) -> torch.Tensor:
    """cv::resize(frame, res, Size(), f, f) with f = 640 / cols (main.cpp:242,287,311): 8UC3 fixed-point bilinear."""
    b = batch_of(frame)
    oh, ow = C.c_int(0), C.c_int(0)
    ctx._l.uwip_overlap_working_size(b.rows, b.cols, C.byref(oh), C.byref(ow))
    shape = (b.frames, oh.value, ow.value, 3) if frame.dim() == 4 else (oh.value, ow.value, 3)
    out = torch.empty(shape, dtype=torch.uint8, device=frame.device)
    ob = batch_of(out)
    torch.cuda.current_stream(frame.device).synchronize()
    ctx.call("uwip_resize_bgr", C.byref(b), C.byref(ob))
    ctx.sync()
    return out


def calcBlur(ctx: Context, frame: torch.Tensor):
    """videostrip.cpp:170-184 per BGR frame -> float (or a tensor for a batch)."""
    b = batch_of(frame)
    out = torch.empty((b.frames,), dtype=torch.float32, device=frame.device)
    torch.cuda.current_stream(frame.device).synchronize()
    ctx.call("uwip_calcBlur", C.byref(b), C.c_void_p(out.data_ptr()))
    ctx.sync()
    return float(out.cpu()[0]) if frame.dim() == 3 else out


def select_keyframes(ctx: Context, frames, minOverlap: float = OVERLAP_MIN, kWindow: int = DEFAULT_KWINDOW, seed: int = 1,
                     report=None):
    """The selector loop of src/main.cpp:284-394 over a sequence of full-resolution BGR frames (CUDA uint8
    tensors): first frame = key frame; a frame whose overlap with the key frame is <= minOverlap (or that
    yields -2.0 -> OVERLAP_MIN + 0.01, :321-326) triggers the sharpest-of-the-next-k refinement (:335-366).
    Returns the exported rows [(id, frame_number, overlap, blur)], the same columns as the reference's TSV
    report (:263,297,381).  Ends when the frames run out (the reference exits there, B-14)."""
    global videoWidth, videoHeight
    frames = list(frames)
    if not frames:
        return []
    videoHeight, videoWidth = int(frames[0].shape[0]), int(frames[0].shape[1])
    h = int(round(videoHeight * (TARGET_WIDTH / videoWidth)))
    w = int(round(videoWidth * (TARGET_WIDTH / videoWidth)))

    def resized(f):          # cv::resize(frame, res_frame, Size(), f, f), main.cpp:311: what calcBlur receives (:338,355)
        return resize_bgr(ctx, f)

    rows = [(0, 0, 0.0, 0.0)]
    kf = keyframe(ctx, frames[0])
    nxt, read = 1, 1
    while nxt < len(frames):
        f = frames[nxt]; nxt += 1; read += 1
        ov = calcOverlap(ctx, kf, f, seed)
        if ov == -2.0:
            ov = OVERLAP_MIN + 0.01
        if ov <= minOverlap:
            best, bestn, bf = calcBlur(ctx, resized(f)), nxt - 1, f
            eof = False
            for _ in range(kWindow):
                if nxt >= len(frames):
                    eof = True
                    break
                g = frames[nxt]; nxt += 1; read += 1
                b = calcBlur(ctx, resized(g))
                if b > best:
                    best, bestn, bf = b, read, g
            kf.feats.close()
            kf = keyframe(ctx, bf)
            rows.append((len(rows), bestn, float(ov), float(best)))
            if report is not None:
                report.write(f"{rows[-1][0]}\t{bestn}\t\t{ov}\t{best}\n")
            if eof:
                break
    kf.feats.close()
    return rows
