"""The host-buffer variant of the bench (4 sub-batch threads x upload -> pipe -> download), a few steps, meant to run
under  rocprofv3 --kernel-trace --memory-copy-trace  (GPU box).  `tools/host_trace_digest.py` turns the two CSVs into
the timeline summary kept under profiles/.
    python3 tools/host_trace.py [steps] [variant]
variant: "bench" = exactly bench.py's host leg (run_host with prefetch)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uwimageproc_amd import synth
from uwimageproc_amd.pipeline import FramePipe

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
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


go(1)
t0 = time.perf_counter(); go(K); dt = time.perf_counter() - t0
print(f"host mode: {S*Fs*K/dt:8.1f} frames/s  {dt/K*1e3:6.1f} ms per step over {K} steps", flush=True)
