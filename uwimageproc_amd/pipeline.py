"""The per-frame pipe  bgdehaze -> histretch -> aclahe -> videostrip-overlap  on a
batch of frames resident in HBM, driven through the C ABI with one stream and no
intermediate host synchronisation except where the reference's algorithm has a
host decision (the ACLAHE parameter choice)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import aclahe
from ._native import Context, batch_of

DEHAZE_FULL, DEHAZE_GUARD_S = 1, 2


class FramePipe:
    def __init__(self, device: int, frames: int, rows: int, cols: int, letters: str = "RGB", w: int = 15):
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        # share torch's current stream so torch events / synchronize cover our kernels
        self.ctx = Context(device, stream=torch.cuda.current_stream(self.dev).cuda_stream)
        self.F, self.H, self.W = frames, rows, cols
        self.letters, self.w = letters.encode(), w
        self.work = torch.empty((frames, rows, cols, 3), dtype=torch.uint8, device=self.dev)
        self.v = torch.empty((frames, rows, cols), dtype=torch.uint8, device=self.dev)
        self.v_out = torch.empty_like(self.v)
        self.params = []
        self.h_bs = (C.c_int32 * frames)()
        self.h_cl = (C.c_int32 * frames)()

    def stage_dehaze(self, src: torch.Tensor):
        sb, ob = batch_of(src), batch_of(self.work)
        self.ctx.call("uwip_dehaze", C.byref(sb), C.byref(ob), self.w, DEHAZE_FULL | DEHAZE_GUARD_S, None, None, None)

    def stage_histretch(self):
        b = batch_of(self.work)
        self.ctx.call("uwip_histretch", C.byref(b), self.letters, 2, 98)

    def stage_aclahe(self):
        wb, vb, ob = batch_of(self.work), batch_of(self.v), batch_of(self.v_out)
        self.ctx.call("uwip_bgr_to_v", C.byref(wb), C.byref(vb))
        # sweep -> host parameter choice (ACLAHE.py:66-129, native MINPACK restatement) -> per-frame CLAHE
        self.ctx.call("uwip_aclahe_auto", C.byref(vb), C.byref(ob), 0, self.h_bs, self.h_cl)
        self.params = list(zip(self.h_bs, self.h_cl))

    def stages(self):
        return ["bgdehaze(adaptiveExp_map,w=15)", "histretch(RGB,2/98)", "aclahe(sweep+select+apply on V)"]

    def run(self, src: torch.Tensor):
        self.stage_dehaze(src)
        self.stage_histretch()
        self.stage_aclahe()
        return self.v_out
