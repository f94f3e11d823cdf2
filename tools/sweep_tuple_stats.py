"""How much exact work the ACLAHE sweep has at (cell, grey level) granularity, on the bench's frames (GPU box).
k_clahe_sweep skips a clip limit whose LUTs repeat the previous limit's for the whole cell.  Finer: for ONE grey level v of
a cell the four LUT entries (L00[v], L01[v], L10[v], L11[v]) repeat across limits far more often, and where the four agree
the blend is that value whatever the position.  Prints, per grid, evaluations per pixel (of 51):
  all limits | limits with LUTs of their own (what the kernel pays) | distinct 4-tuples per (cell, v) | ... that are not flat."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from uwimageproc_amd import aclahe, synth
from uwimageproc_amd.pipeline import FramePipe

F, H, W = 2, 1080, 1920
pipe = FramePipe(0, F, H, W, guard_s=True)
src = torch.from_numpy(synth.uw_stream_motion(0, F, H, W) if hasattr(synth, "uw_stream_motion") else synth.uw_stream(0, F, H, W)).cuda()
pipe.stage_dehaze(src); pipe.stage_histretch()
v = aclahe.GaussianBlur3(pipe.ctx, aclahe.bgr_to_v(pipe.ctx, pipe.work)).cpu().numpy()
limits = np.arange(0, 25.5, 0.5)


def luts_of(hist, clip, area):
    """hist [T,256] int64 -> LUT [T,256] uint8 (OpenCV's clip / redistribute / cumulative rule, SURVEY A-3)"""
    h = hist.copy()
    if clip > 0:
        clipped = np.maximum(h - clip, 0).sum(axis=1)
        h = np.minimum(h, clip)
        batch = clipped // 256
        resid = clipped - batch * 256
        h += batch[:, None]
        for t in np.nonzero(resid)[0]:
            step = max(256 // int(resid[t]), 1)
            idx = np.arange(0, 256, step)[: int(resid[t])]
            h[t, idx] += 1
    scale = np.float32(255.0) / np.float32(area)
    return np.clip(np.rint(np.cumsum(h, axis=1).astype(np.float32) * scale), 0, 255).astype(np.uint8)


for g in (2, 4, 8, 16, 32):
    tw, th = -(-W // g), -(-H // g)
    pw, ph = tw * g, th * g
    area = tw * th
    clips = [0 if c == 0 else max(int(c * area / 256), 1) for c in limits]
    tot = np.zeros(4)
    for f in range(F):
        p = np.pad(v[f], ((0, ph - H), (0, pw - W)), mode="reflect") if (pw, ph) != (W, H) else v[f]
        tiles = p.reshape(g, th, g, tw).transpose(0, 2, 1, 3).reshape(g * g, -1)
        hist = np.stack([np.bincount(r, minlength=256) for r in tiles]).astype(np.int64)
        L = np.stack([luts_of(hist, c, area) for c in clips]).reshape(51, g, g, 256)        # [limit][ty][tx][v]
        # interpolation cells: pixel (x, y) blends tiles floor((x / tw) - 0.5), +1 (clamped)
        ys, xs = np.arange(H), np.arange(W)
        cy = np.floor(ys / th - 0.5).astype(int) + 1          # 0 .. g
        cx = np.floor(xs / tw - 0.5).astype(int) + 1
        cell = cy[:, None] * (g + 1) + cx[None, :]
        cnt = np.zeros(((g + 1) * (g + 1), 256), np.int64)
        np.add.at(cnt, (cell.ravel(), v[f].ravel()), 1)
        cnt = cnt.reshape(g + 1, g + 1, 256)
        for iy in range(g + 1):
            y0, y1 = max(iy - 1, 0), min(iy, g - 1)
            for ix in range(g + 1):
                x0, x1 = max(ix - 1, 0), min(ix, g - 1)
                c = cnt[iy, ix]
                n = c.sum()
                if n == 0:
                    continue
                tup = np.stack([L[:, y0, x0], L[:, y0, x1], L[:, y1, x0], L[:, y1, x1]], axis=2).astype(np.uint32)  # [51][256][4]
                key = (tup[..., 0] << 24) | (tup[..., 1] << 16) | (tup[..., 2] << 8) | tup[..., 3]                   # [51][256]
                own_lut = 1 + (np.diff(key, axis=0) != 0).any(axis=1).sum()          # limits whose LUT set differs from the previous
                ks = np.sort(key, axis=0)
                newk = np.concatenate([np.ones((1, 256), bool), np.diff(ks, axis=0) != 0], axis=0)
                distinct = newk.sum(axis=0)                                             # per v
                flat = ((ks >> 24) == ((ks >> 16) & 255)) & ((ks >> 24) == ((ks >> 8) & 255)) & ((ks >> 24) == (ks & 255))
                nonflat = (newk & ~flat).sum(axis=0)
                tot += np.array([51.0 * n, float(own_lut) * n, float((c * distinct).sum()), float((c * nonflat).sum())])
    per_px = tot / (F * H * W)
    print(f"grid {g:2d}: evaluations per pixel  all {per_px[0]:5.1f}  own-LUT limits {per_px[1]:5.1f}  distinct tuples {per_px[2]:5.1f}  "
          f"non-flat distinct tuples {per_px[3]:5.1f}", flush=True)
