"""k_clahe_tilehist alone on pipeline-like V planes: per-launch time by grid (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import uwimageproc_amd as uw
from uwimageproc_amd import aclahe, synth, batch_of
from uwimageproc_amd.pipeline import FramePipe
F, H, W = 64, 1080, 1920
pipe = FramePipe(0, F, H, W, guard_s=True)
base = synth.uw_stream(0, 16, H, W)
src = torch.from_numpy(np.concatenate([base] * 4)).cuda()
pipe.stage_dehaze(src); pipe.stage_histretch()
v = aclahe.bgr_to_v(pipe.ctx, pipe.work)
if os.environ.get("RANDOM_V"):
    v = torch.randint(0, 256, v.shape, dtype=torch.uint8, device=v.device)
out = torch.empty_like(v)
ctx = pipe.ctx
for g in (2, 4, 8, 16, 32):
    ib, ob = batch_of(v), batch_of(out)
    ctx.call("uwip_clahe", C.byref(ib), C.byref(ob), C.c_double(3.0), g, g, 0)
    ctx.sync()
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(3):
        ctx.call("uwip_clahe", C.byref(ib), C.byref(ob), C.c_double(3.0), g, g, 0)
    ctx.sync()
    r = ctx.prof_results(); ctx.prof_enable(False)
    print(f"grid {g:2d}: " + "  ".join(f"{k} {ms/cnt*1e3:7.1f} us" for k, (ms, cnt) in r.items()) + f"   whole {sum(ms/cnt for ms, cnt in r.values())*1e3:7.1f} us = {3.0*H*W*F/(sum(ms/cnt for ms, cnt in r.values())*1e-3)/8e12:.3f} of 8 TB/s", flush=True)
