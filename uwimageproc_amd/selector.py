"""The key-frame selector of modules/videostrip/src/main.cpp:300-394 across the GPUs of a node (SURVEY.md section 8e,
option 1; the reference's own multi-device note is a TODO, main.cpp:216).

What is sequential in the selector is only the DECISION chain: each frame is compared with the current key frame, and
which frame is the key frame depends on the earlier decisions.  What costs time -- resize, detect, describe, the blur
metric -- depends on one frame only.  So:
  1. every rank takes a contiguous slice of the frames (sharding.frame_slice) and extracts, per frame, the cached part
     of `struct keyframe` (keypoints + descriptors, videostrip.hpp:62-68) and the calcBlur value -- no halo, no
     data-path collective;
  2. the per-frame records (<= 196 KB + one float) travel to rank 0 over the control plane;
  3. rank 0 replays main.cpp:300-394 on them: threshold test, -2.0 -> 0.41, sharpest-of-the-next-k refinement.  The
     overlaps of the next `lookahead` frames against the current key frame are evaluated in ONE matcher launch
     (speculatively: those after a trigger are discarded), so the chain is not launch-latency bound.
The result is exactly the single-GPU selector's: features are a deterministic function of the frame.

The compute back end is injected (`Backend`): the GPU one below drives the C ABI; the CPU tests inject the oracle."""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np

from . import sharding

OVERLAP_MIN = 0.4          # videostrip.hpp:50
DEFAULT_KWINDOW = 11       # videostrip.hpp:51


class Backend:
    """extract(frames [B,H,W,3] uint8) -> list of (kps, desc, blur) host records, one per frame;
    overlaps(key_record, [records]) -> list of float ratios (calcOverlap of each record against the key)."""

    def extract(self, frames: np.ndarray):
        raise NotImplementedError

    def overlaps(self, key, objs: Sequence) -> List[float]:
        raise NotImplementedError


def chain(records: Sequence, overlaps: Callable, minOverlap: float = OVERLAP_MIN, kWindow: int = DEFAULT_KWINDOW,
          lookahead: int = 8) -> List[Tuple[int, int, float, float]]:
    """main.cpp:284-394 on per-frame records (kps, desc, blur): returns the report rows (ID, Frame, Overlap, Blur), the
    same columns as videostrip.select_keyframes / the CLI's TSV.  `Frame` keeps the reference's numbering: the
    trigger frame's 0-based index (:340 reads it before the window), a window frame's 1-based count of frames read."""
    n = len(records)
    if n == 0:
        return []
    rows = [(0, 0, 0.0, 0.0)]
    key = 0
    nxt, read = 1, 1
    spec_key, spec_at, spec = -1, 0, []                 # overlaps of frames spec_at.. against frame spec_key
    while nxt < n:
        if spec_key != key or not (spec_at <= nxt < spec_at + len(spec)):
            idx = list(range(nxt, min(n, nxt + max(1, lookahead))))
            spec = overlaps(records[key], [records[i] for i in idx])
            spec_key, spec_at = key, nxt
        ov = float(spec[nxt - spec_at])
        cur = nxt
        nxt += 1; read += 1
        if ov == -2.0:
            ov = OVERLAP_MIN + 0.01                     # :321-326
        if ov <= minOverlap:                            # :329
            best, bestn, bi = float(records[cur][2]), nxt - 1, cur
            eof = False
            for _ in range(kWindow):                    # :344-366
                if nxt >= n:
                    eof = True
                    break
                g = nxt
                nxt += 1; read += 1
                b = float(records[g][2])
                if b > best:
                    best, bestn, bi = b, read, g
            key = bi
            rows.append((len(rows), bestn, ov, best))
            if eof:
                break
    return rows


def select_distributed(backend: Backend, read_frame: Callable[[int], np.ndarray], n_frames: int, rank: int = 0, world: int = 1,
                       minOverlap: float = OVERLAP_MIN, kWindow: int = DEFAULT_KWINDOW, batch: int = 8, lookahead: int = 8,
                       group=None):
    """Every rank extracts its slice; rank 0 runs the chain; every rank returns the rows (broadcast).  With world == 1
    (or torch.distributed not initialised) it is the single-GPU selector on precomputed records."""
    a, b = sharding.frame_slice(n_frames, rank, world)
    local = []
    for k in range(a, b, batch):
        idx = list(range(k, min(b, k + batch)))
        local += list(backend.extract(np.stack([read_frame(i) for i in idx])))
    if world == 1:
        return chain(local, backend.overlaps, minOverlap, kWindow, lookahead)
    import torch.distributed as dist
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0, group=group)          # control plane: <= 196 KB per frame
    rows = None
    if rank == 0:
        records = [r for part in gathered for r in part]             # slices are contiguous and in rank order
        assert len(records) == n_frames
        rows = chain(records, backend.overlaps, minOverlap, kWindow, lookahead)
    box = [rows]
    dist.broadcast_object_list(box, src=0, group=group)
    return box[0]


class GpuBackend(Backend):
    """The C ABI: uwip_overlap_detect + uwip_features_download / uwip_calcBlur(uwip_resize_bgr) per rank;
    uwip_features_upload + uwip_overlap_match on rank 0.  videoWidth / videoHeight are the reference's globals
    (main.cpp:238-239): the full-resolution size as written (SURVEY B-8)."""

    def __init__(self, ctx, video_size, seed: int = 1, lookahead: int = 8):
        from . import videostrip as vs
        self.vs, self.ctx, self.seed = vs, ctx, seed
        self.vw, self.vh = video_size
        self.feats = vs.Features(ctx, 1 + max(1, lookahead))
        self.cap = max(1, lookahead)
        self.key = None
        self.wh = None

    def close(self):
        self.feats.close()

    def extract(self, frames: np.ndarray):
        import torch
        vs = self.vs
        t = torch.from_numpy(np.ascontiguousarray(frames)).cuda(self.ctx.device)
        f = vs.Features(self.ctx, len(frames))
        f.detect(t, 0)
        blur = vs.calcBlur(self.ctx, vs.resize_bgr(self.ctx, t)).cpu().numpy()
        out = []
        for s in range(len(frames)):
            kps, desc = f.download(s)
            out.append((kps, desc, float(blur[s])))
        f.close()
        return out

    def _upload(self, slot, rec):
        import ctypes as C
        if self.wh is None:
            oh, ow = C.c_int(0), C.c_int(0)
            self.ctx._l.uwip_overlap_working_size(int(self.vh), int(self.vw), C.byref(oh), C.byref(ow))
            self.wh = (oh.value, ow.value)
        kps, desc = np.ascontiguousarray(rec[0]), np.ascontiguousarray(rec[1])
        self.ctx.call("uwip_features_upload", self.feats._h, int(slot), self.wh[0], self.wh[1], C.c_void_p(kps.ctypes.data),
                      C.c_void_p(desc.ctypes.data), len(kps))

    def overlaps(self, key, objs):
        if self.key is not key:                              # the key frame's features are uploaded once per key change
            self._upload(0, key)
            self.key = key                                   # keeps the record alive, so identity is a sound test
        out = []
        for k in range(0, len(objs), self.cap):
            part = objs[k:k + self.cap]
            for j, rec in enumerate(part):
                self._upload(1 + j, rec)
            r = self.vs.match_pairs(self.ctx, self.feats, self.feats, list(range(1, 1 + len(part))), [0] * len(part), self.vw, self.vh,
                                    self.seed)
            out += [float(v) for v in r["ratio"].cpu().numpy()]
        return out
