import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np, ctypes as C
import uwimageproc_amd as uw
from uwimageproc_amd import aclahe, batch_of
ctx = uw.Context(0)
for (F,H,W) in ((16,2160,3840),(64,1080,1920)):
    v = torch.randint(0,256,(F,H,W),dtype=torch.uint8,device='cuda'); o = torch.empty_like(v)
    for g in (2,8,16,32):
        c = aclahe.CLAHE(ctx, 3.0, (g,g)); c.apply(v,o)
        ctx.prof_reset(); ctx.prof_enable(True)
        for _ in range(5): c.apply(v,o)
        ctx.sync(); r = ctx.prof_results(); ctx.prof_enable(False)
        ms, n = r['k_clahe_apply']; ms/=n
        print(F,H,W,'g',g,'apply %.1f us  %.1f%% of HBM'%(ms*1e3, 2*F*H*W/(ms*1e-3)/8e12*100))
