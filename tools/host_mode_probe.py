"""Where the host-buffer variant of the bench loses time: the same 4 sub-batch threads with no copies / uploads only /
downloads only / both (GPU box)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uwimageproc_amd import synth
from uwimageproc_amd.pipeline import FramePipe
S, Fs, H, W, K = 4, 64, 1080, 1920, 5
frames = synth.uw_stream(0, 16, H, W)
pipes, bufs, srcs = [], [], []
for i in range(S):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):                 # the pipe's context takes torch's current stream
        p = FramePipe(0, Fs, H, W)
    pipes.append(p)
    hb = p.host_buffers(); hb[0][...] = np.concatenate([frames] * (Fs // 16)); bufs.append(hb)
    p.src_dev = torch.from_numpy(np.ascontiguousarray(hb[0])).cuda()
torch.cuda.synchronize()
def variant(up, down):
    def loop(i):
        p = pipes[i]
        for _ in range(K):
            if up: p.ctx.h2d_async(p.src_dev, bufs[i][0])
            p.run(p.src_dev)
            if down: p.ctx.d2h_async(bufs[i][1], p.work)
    def go():
        th = [threading.Thread(target=loop, args=(i,)) for i in range(S)]
        [t.start() for t in th]; [t.join() for t in th]
        for p in pipes: p.ctx.sync()
        torch.cuda.synchronize()
    go()
    t0 = time.perf_counter(); go(); dt = time.perf_counter() - t0
    print(f"upload={up} download={down}: {S*Fs*K/dt:8.1f} frames/s  {dt/K*1e3:6.1f} ms per step", flush=True)
for up, down in ((0, 0), (1, 0), (0, 1), (1, 1)):
    variant(up, down)
