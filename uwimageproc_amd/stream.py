"""A finite frame stream across the ranks of one node: one process per GPU, contiguous frame
slices (sharding.frame_slice), NO data-path collective.

bgdehaze, histretch and aclahe depend on one frame only.  The overlap ratio of frame i (calcOverlap against its
predecessor, videostrip.cpp:192-289 as the pipe chains it) needs the PROCESSED frame i-1, which for a rank's first frame
belongs to the previous rank.  Seam rule: a rank with start > 0 also runs frame start-1 through the pipe (a one-frame
halo) and discards that frame's own outputs -- every reported value is then exactly what a single GPU computes for the
whole stream; the cost is one extra frame per rank.  The only exchange is the control-plane gather of the per-frame
scalars (ratio, BS, CL) into frame order.

The reference's multi-device note is a TODO (modules/videostrip/src/main.cpp:216); its sequential key-frame chain
(:300-394) stays on one rank: `select_from_ratios` below replays the threshold test on the gathered per-frame values
only for the predecessor-chained variant the pipe measures, not the key-frame-chained selector of the CLI."""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np

from . import sharding


class StreamDriver:
    """process(batch [F,H,W,3] uint8, first_of_stream: bool) -> (out [F,H,W,3] uint8, ratio [F], params [F] of (BS, CL));
    it must chain the overlap across consecutive calls (FramePipe does)."""

    def __init__(self, n_frames: int, rank: int, world: int, batch: int, process: Callable):
        self.n, self.rank, self.world, self.batch, self.process = n_frames, rank, world, batch, process
        self.start, self.stop = sharding.frame_slice(n_frames, rank, world)
        self.halo = 1 if (self.start > 0 and self.stop > self.start) else 0

    def frame_indices(self) -> List[int]:
        return list(range(self.start - self.halo, self.stop))

    def run(self, read_frame: Callable[[int], np.ndarray], sink: Callable[[int, np.ndarray], None] = None):
        """read_frame(i) -> HxWx3 uint8; sink(i, out_frame) receives this rank's processed frames.  Returns the local
        per-frame lists (ratios, params) for frames [start, stop)."""
        idx = self.frame_indices()
        ratios, params = [], []
        first = True
        for k in range(0, len(idx), self.batch):
            chunk = idx[k:k + self.batch]
            frames = [read_frame(i) for i in chunk]
            valid = len(frames)
            while len(frames) < self.batch:              # pad the last batch; padded outputs are dropped
                frames.append(frames[-1])
            out, ratio, par = self.process(np.stack(frames), first)
            first = False
            for j in range(valid):
                i = chunk[j]
                if i < self.start:                       # the halo frame: only its features were needed
                    continue
                if sink is not None:
                    sink(i, out[j])
                ratios.append(float(ratio[j]))
                params.append((int(par[j][0]), int(par[j][1])))
        return ratios, params

    def gather(self, ratios: Sequence[float], params: Sequence[Tuple[int, int]]):
        """Every rank receives the whole stream's per-frame values in frame order (control plane only)."""
        r = sharding.gather_in_frame_order(list(ratios), self.n, self.rank, self.world)
        bs = sharding.gather_in_frame_order([float(p[0]) for p in params], self.n, self.rank, self.world)
        cl = sharding.gather_in_frame_order([float(p[1]) for p in params], self.n, self.rank, self.world)
        return r, [(int(a), int(b)) for a, b in zip(bs, cl)]


def pipe_process(pipe):
    """Adapter: a FramePipe as the `process` callable of StreamDriver (host frames in, host results out)."""
    import torch

    def proc(batch: np.ndarray, first_of_stream: bool):
        if first_of_stream:
            pipe.have_prev = False
        src = torch.from_numpy(np.ascontiguousarray(batch)).to(pipe.dev)
        out, ratio = pipe.run(src)
        pipe.ctx.sync()
        torch.cuda.synchronize(pipe.dev)
        return out.cpu().numpy(), ratio.cpu().numpy(), list(pipe.params)

    return proc


def select_from_ratios(ratios: Sequence[float], minOverlap: float) -> List[int]:
    """Frames whose overlap with their predecessor is at or below the threshold (the test of main.cpp:329 applied to
    the predecessor chain; -2.0 counts as OVERLAP_MIN + 0.01, :321-326)."""
    out = []
    for i, r in enumerate(ratios):
        v = 0.41 if r == -2.0 else r
        if i > 0 and v <= minOverlap:
            out.append(i)
    return out
