"""Per-kernel time of one 64-frame sub-batch of the 1080p pipe on one stream (HIP events inside libuwip): the quick
before/after table for kernel work (GPU box).   python3 tools/kernel_times.py [frames] [reps]   (KT_4K=1: 3840x2160)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uwimageproc_amd import synth
from uwimageproc_amd.pipeline import FramePipe
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W = (2160, 3840) if os.environ.get("KT_4K") else (1080, 1920)
base = synth.uw_stream(0, min(F, 16), H, W)
src = torch.from_numpy(np.concatenate([base] * ((F + len(base) - 1) // len(base)))[:F]).cuda()
pipe = FramePipe(0, F, H, W, guard_s=True)
pipe.run(src); torch.cuda.synchronize()
pipe.ctx.prof_reset(); pipe.ctx.prof_enable(True)
for _ in range(reps):
    pipe.run(src)
torch.cuda.synchronize()
res = pipe.ctx.prof_results()
tot = sum(ms for ms, _ in res.values()) / reps
print(f"{F} frames of {W}x{H}, {reps} reps: {tot:.3f} ms of kernels per step")
for k, (ms, cnt) in sorted(res.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:26s} {ms/reps:8.3f} ms  {cnt/reps:5.1f} launches  {100*ms/reps/tot:5.1f} %")
