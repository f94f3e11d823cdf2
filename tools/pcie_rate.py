"""Raw host <-> HBM copy rates of the library's page-locked buffers (GPU box): one direction alone, both at once."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uwimageproc_amd as uw
n = 64 * 1080 * 1920 * 3
c1, c2 = uw.Context(0, stream=torch.cuda.Stream().cuda_stream), uw.Context(0, stream=torch.cuda.Stream().cuda_stream)
h1, h2 = c1.host_alloc((n,)), c2.host_alloc((n,))
d1 = torch.empty(n, dtype=torch.uint8, device="cuda"); d2 = torch.empty_like(d1)
def t(fn, reps=5):
    fn(); c1.sync(); c2.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    c1.sync(); c2.sync()
    return (time.perf_counter() - t0) / reps
a = t(lambda: c1.h2d_async(d1, h1)); print("H2D alone   %.1f GB/s" % (n / a / 1e9), flush=True)
b = t(lambda: c2.d2h_async(h2, d2)); print("D2H alone   %.1f GB/s" % (n / b / 1e9), flush=True)
c = t(lambda: (c1.h2d_async(d1, h1), c2.d2h_async(h2, d2))); print("both at once %.1f GB/s each way" % (n / c / 1e9), flush=True)
p = torch.empty(n, dtype=torch.uint8).pin_memory()
e = t(lambda: d1.copy_(p, non_blocking=True)); torch.cuda.synchronize(); print("torch pinned H2D %.1f GB/s" % (n / e / 1e9))
