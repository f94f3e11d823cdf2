"""How many of the sweep's 51 clip limits give distinct LUTs per interpolation cell, on the bench's frames (GPU box).
Distinct limits cost an evaluation per pixel; a limit at or above the tallest bin of the cell's tiles repeats the unclipped LUT."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from uwimageproc_amd import aclahe, synth
from uwimageproc_amd.pipeline import FramePipe
F, H, W = (2, 2160, 3840) if os.environ.get("STATS_4K") else (8, 1080, 1920)
pipe = FramePipe(0, F, H, W, guard_s=True)
src = torch.from_numpy(synth.uw_stream(0, F, H, W)).cuda()
pipe.stage_dehaze(src); pipe.stage_histretch()
v = aclahe.GaussianBlur3(pipe.ctx, aclahe.bgr_to_v(pipe.ctx, pipe.work)).cpu().numpy()
limits = np.arange(0, 25.5, 0.5)
for g in (2, 4, 8, 16, 32):
    tw, th = -(-W // g), -(-H // g)
    pw, ph = tw * g, th * g
    area = tw * th
    clips = np.array([0 if c == 0 else max(int(c * area / 256), 1) for c in limits])
    frac = []
    for f in range(F):
        p = np.pad(v[f], ((0, ph - H), (0, pw - W)), mode="reflect")
        t = p.reshape(g, th, g, tw).transpose(0, 2, 1, 3).reshape(g * g, -1)
        mx = np.array([np.bincount(r, minlength=256).max() for r in t]).reshape(g, g)
        mp = np.pad(mx, 1, mode="edge")
        cellmax = np.maximum(np.maximum(mp[:-1, :-1], mp[:-1, 1:]), np.maximum(mp[1:, :-1], mp[1:, 1:]))   # (g+1, g+1) cells
        for grp in range(3):
            cl = clips[grp * 17:(grp + 1) * 17]
            # distinct LUTs within the group: consecutive limits differ when their integer clips differ and the lower one still clips
            eff = np.minimum(cl[None, None, :], cellmax[:, :, None])
            eff = np.where(cl[None, None, :] == 0, cellmax[:, :, None], eff)
            distinct = 1 + (np.diff(eff, axis=2) != 0).sum(axis=2)
            frac.append(distinct.mean() / 17.0)
    print(f"grid {g:2d}: tile area {area:7d}  clip(25) {clips[-1]:6d}  tallest bin median {int(np.median(mx))}  "
          f"mean fraction of the 17 limits with their own LUT: {np.mean(frac):.2f} (groups {np.mean(frac[0::3]):.2f} {np.mean(frac[1::3]):.2f} {np.mean(frac[2::3]):.2f})", flush=True)
