"""A/B of the sweep's remainder handling in one process is not possible (the mode is read once): run this tool once per
mode, several times, on the same box:  UWIP_SWEEP_REM=0|1|2 python3 tools/sweep_ab.py   -> ms per 64-frame sweep"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uwimageproc_amd import aclahe, synth
from uwimageproc_amd.pipeline import FramePipe
F, H, W = 64, 1080, 1920
pipe = FramePipe(0, F, H, W, guard_s=True)
base = synth.uw_stream_motion(0, 16, H, W) if os.environ.get("MOTION") else synth.uw_stream(0, 16, H, W)
src = torch.from_numpy(np.concatenate([base] * 4)).cuda()
pipe.stage_dehaze(src); pipe.stage_histretch()
v = aclahe.GaussianBlur3(pipe.ctx, aclahe.bgr_to_v(pipe.ctx, pipe.work), 0)
ctx = pipe.ctx
aclahe.sweep(ctx, v); ctx.sync()
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(5):
    aclahe.sweep(ctx, v)
ctx.sync()
ms, cnt = ctx.prof_results()["k_clahe_sweep"]
print(f"rem {os.environ.get('UWIP_SWEEP_REM', '1')} int {os.environ.get('UWIP_SWEEP_INT', '1')} motion {os.environ.get('MOTION', '')}: {ms / 5:.3f} ms per 64-frame sweep", flush=True)
