"""Event timeline of the host-buffer mode WITHOUT a profiler attached (GPU box): per sub-batch and step, when the stage
groups start and end on the device clock, and how long each sub-batch stream sat idle between two batches.   python3 tools/host_timeline.py [steps]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uwimageproc_amd import synth
from uwimageproc_amd.pipeline import FramePipe

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S, Fs, H, W = 4, 64, 1080, 1920
frames = synth.uw_stream(0, 16, H, W)
from uwimageproc_amd import Copier
copier = Copier(0)
pipes, bufs = [], []
for i in range(S):
    with torch.cuda.stream(torch.cuda.Stream()):
        p = FramePipe(0, Fs, H, W, copier=copier)
    pipes.append(p)
    hb = p.host_buffers()
    hb[0][...] = np.concatenate([frames] * (Fs // 16))
    bufs.append(hb)
torch.cuda.synchronize()


def go(k):
    def loop(i):
        for _ in range(k):
            pipes[i].run_host(bufs[i][0], bufs[i][1], prefetch=bufs[i][0])
    th = [threading.Thread(target=loop, args=(i,)) for i in range(S)]
    [t.start() for t in th]; [t.join() for t in th]
    for p in pipes:
        p.sync()
    torch.cuda.synchronize()


go(2)
base = torch.cuda.Event(enable_timing=True); base.record(); torch.cuda.synchronize()
for p in pipes:
    p.timeline = []
t0 = time.perf_counter(); go(K); dt = time.perf_counter() - t0
print(f"host mode: {S*Fs*K/dt:8.1f} frames/s  {dt/K*1e3:6.1f} ms per step over {K} steps")
rows = []
for i, p in enumerate(pipes):
    ev = {(k, name): base.elapsed_time(e) for name, k, e in p.timeline}
    for k in sorted({k for k, _ in ev}):
        g = lambda n: ev.get((k, n), float("nan"))
        rows.append((g("dehaze0"), i, k, g("dehaze1"), g("aclahe1"), g("overlap1")))
print("sub step | dehaze0 dehaze1 aclahe1 overlap1 | gap between this batch's first kernel and the previous batch's last (ms)")
last, tot = {}, 0.0
for d0, i, k, d1, a1, o1 in sorted(rows):
    stall = d0 - last.get(i, d0)
    tot += stall
    last[i] = o1
    print(f"{i:3d} {k:4d} | {d0:7.1f} {d1:7.1f} {a1:7.1f} {o1:7.1f} | {stall:6.2f}")
print(f"sum of gaps {tot:.1f} ms over {K} steps x {S} sub-batches")
