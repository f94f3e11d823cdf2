"""Does it pay to keep the four sub-batches out of phase (one in the sweep while another is in the guided filter)?
Resident frames, 20 steps, sub-batch thread i starts i * delta late (GPU box)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uwimageproc_amd import synth
from uwimageproc_amd.pipeline import FramePipe
S, Fs, H, W, K = 4, 64, 1080, 1920, 20
frames = synth.uw_stream(0, 16, H, W)
pipes, srcs = [], []
for i in range(S):
    with torch.cuda.stream(torch.cuda.Stream()):
        pipes.append(FramePipe(0, Fs, H, W))
    srcs.append(torch.from_numpy(np.concatenate([frames] * (Fs // 16))).cuda())
torch.cuda.synchronize()
def go(delta, k):
    def loop(i):
        time.sleep(i * delta)
        for _ in range(k):
            pipes[i].run(srcs[i])
    th = [threading.Thread(target=loop, args=(i,)) for i in range(S)]
    [t.start() for t in th]; [t.join() for t in th]
    for p in pipes: p.ctx.sync()
    torch.cuda.synchronize()
go(0.0, 2)
for delta in (0.0, 0.012, 0.025, 0.05):
    t0 = time.perf_counter(); go(delta, K); dt = time.perf_counter() - t0
    print(f"delta {delta*1e3:5.1f} ms: {S*Fs*K/dt:8.1f} frames/s ({dt/K*1e3:6.1f} ms per step incl. the ramp)", flush=True)
