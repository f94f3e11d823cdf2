"""Per-kernel timing of the hot-path stages on synthetic frames (hipEvent
profiling inside libuwip).  Usage: python tools/microbench.py [frames] [rows] [cols]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import uwimageproc_amd as uw
from uwimageproc_amd import aclahe, preprocessing as pp, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
ctx = uw.Context(0)
base = synth.uw_batch(0, 4, rows, cols)
frames = torch.from_numpy(np.concatenate([base] * (F // 4), axis=0)).cuda()
N = rows * cols
print(f"frames={F} {cols}x{rows}  batch={frames.numel()/1e6:.1f} MB")


def run(name, fn, reps=5):
    fn()
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.time()
    for _ in range(reps):
        fn()
    ctx.sync()
    wall = (time.time() - t0) / reps
    res = ctx.prof_results()
    ctx.prof_enable(False)
    print(f"== {name}: wall {wall*1e3:.3f} ms/batch  ({F/wall:.0f} fps)")
    for k, (ms, cnt) in sorted(res.items(), key=lambda kv: -kv[1][0]):
        print(f"   {k:24s} {ms/reps:9.3f} ms/batch  launches/batch={cnt/reps:.0f}")
    return res


work = frames.clone()
run("histretch RGB", lambda: pp.histretch(ctx, work, "RGB"))
r = ctx.prof_results()
v = aclahe.bgr_to_v(ctx, frames)
run("bgr_to_v", lambda: aclahe.bgr_to_v(ctx, frames))
for g in (8, 32):
    c = aclahe.CLAHE(ctx, 3.0, (g, g))
    dst = torch.empty_like(v)
    res = run(f"clahe g={g}", lambda: c.apply(v, dst))
    ms = res["k_clahe_apply"][0] / res["k_clahe_apply"][1]
    print(f"   k_clahe_apply: {2*N*F/ms/1e6:.1f} GB/s algorithmic (2N per frame)")
run("sweep", lambda: aclahe.sweep(ctx, v), reps=2)
