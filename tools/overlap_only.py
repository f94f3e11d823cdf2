"""Only the overlap stage (detect + describe + match) on 64 processed-like 1080p frames, for rocprofv3 --kernel-trace (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uwimageproc_amd import synth
from uwimageproc_amd.pipeline import FramePipe
F, H, W = 64, 1080, 1920
base = synth.uw_stream(0, 16, H, W)
pipe = FramePipe(0, F, H, W, guard_s=True)
pipe.work.copy_(torch.from_numpy(np.concatenate([base] * 4)).cuda())
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    pipe.stage_overlap()
torch.cuda.synchronize()
print("done")
