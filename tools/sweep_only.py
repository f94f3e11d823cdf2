"""Run only the ACLAHE sweep on pipeline-like data (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uwimageproc_amd import aclahe, synth
from uwimageproc_amd.pipeline import FramePipe
F, H, W = 32, 1080, 1920
pipe = FramePipe(0, F, H, W, guard_s=True)
import numpy as np
base = synth.uw_stream(0, 8, H, W)
src = torch.from_numpy(np.concatenate([base] * 4)).cuda()
pipe.stage_dehaze(src); pipe.stage_histretch()
v = aclahe.bgr_to_v(pipe.ctx, pipe.work)
if os.environ.get("SWEEP_RANDOM"):
    v = torch.randint(0, 256, v.shape, dtype=torch.uint8, device=v.device)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    aclahe.sweep(pipe.ctx, v)
print("done")
