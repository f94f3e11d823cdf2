"""The per-frame pipe  bgdehaze -> histretch -> aclahe -> videostrip-overlap  on a
batch of frames resident in HBM, driven through the C ABI on one stream.  The only
host synchronisation inside a step is where the reference's algorithm has a host
decision: the ACLAHE parameter choice (ACLAHE.py:66-129).

Stage definitions (DESIGN.md "the pipe"):
  bgdehaze   main.py:14-20 / adaptiveExp_map (w = 15), S guarded (B-11)
  histretch  -c=RGB, percentiles 2/98 (histretch.cpp:154,217-254)
  aclahe     V of HSV (aclahe.cpp:152-154) -> ParametrosACLAHE (ACLAHE.py:9-129: 3x3 Gaussian prefilter :15,
             sweep = aclahe.cpp:160-193, parameter choice :66-129) -> CLAHE(CL,(BS,BS)) on the unfiltered V
             (python/main.py:19-20) -> back to BGR (aclahe.cpp:216)
  overlap    calcOverlap of every frame against its predecessor (videostrip.cpp:192-289),
             the last frame's features carried into the next batch
"""
from __future__ import annotations

import ctypes as C

import torch

from ._native import Context, Copier, batch_of

DEHAZE_FULL, DEHAZE_GUARD_S = 1, 2
ACLAHE_PREFILTER, ACLAHE_ASYNC = 1, 4          # uwip.h


class FramePipe:
    def __init__(self, device: int, frames: int, rows: int, cols: int, letters: str = "RGB", w: int = 15,
                 video_size=None, seed: int = 1, copier: "Copier | None" = None):
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        # share torch's current stream so torch events / synchronize cover our kernels
        self.stream = torch.cuda.current_stream(self.dev)
        self.ctx = Context(device, stream=self.stream.cuda_stream)
        self.host = None
        self.timeline = None
        self.copier, self._own_copier = copier, False
        self._host_bufs = []
        self.F, self.H, self.W = frames, rows, cols
        self.letters, self.w, self.seed = letters.encode(), w, seed
        # the reference's globals videoWidth / videoHeight (main.cpp:238-239); as written they are the
        # full-resolution size
        self.vw, self.vh = video_size if video_size else (cols, rows)
        self.work = torch.empty((frames, rows, cols, 3), dtype=torch.uint8, device=self.dev)
        self.v = torch.empty((frames, rows, cols), dtype=torch.uint8, device=self.dev)
        self.v_out = torch.empty_like(self.v)
        self.ratio = torch.empty((frames,), dtype=torch.float32, device=self.dev)
        self.info = torch.zeros((frames, 8), dtype=torch.int32, device=self.dev)
        self.h_bs = (C.c_int32 * frames)()
        self.h_cl = (C.c_int32 * frames)()
        self._params = None
        self._inflight = []
        # feature slots: 0 = previous batch's last frame, 1..F = this batch
        fh = C.c_void_p()
        self.ctx.call("uwip_features_create", frames + 1, C.byref(fh))
        self.feats = fh
        self.have_prev = False
        self.pair_q = (C.c_int32 * frames)(*[i + 1 for i in range(frames)])
        self.pair_t = (C.c_int32 * frames)(*[i for i in range(frames)])

    def close(self):
        if getattr(self, "host", None) is not None:
            self.sync()
            self.host = None
        if getattr(self, "_own_copier", False):
            self.copier.close()
            self.copier, self._own_copier = None, False
        for a in getattr(self, "_host_bufs", []):
            if a.ctypes.data in getattr(self.ctx, "_pinned", {}):
                self.ctx.host_free(a)
        self._host_bufs = []
        if getattr(self, "feats", None):
            self.ctx._l.uwip_features_destroy(self.feats)
            self.feats = None

    # ---- stages -----------------------------------------------------------------
    def stage_dehaze(self, src: torch.Tensor):
        sb, ob = batch_of(src), batch_of(self.work)
        self.ctx.call("uwip_dehaze", C.byref(sb), C.byref(ob), self.w, DEHAZE_FULL | DEHAZE_GUARD_S, None, None, None)

    def stage_histretch(self):
        b = batch_of(self.work)
        self.ctx.call("uwip_histretch", C.byref(b), self.letters, 2, 98)

    def stage_dehaze_histretch(self, src: torch.Tensor):
        """The two stages as one chained call: the dehaze writer hands the stretch its histogram."""
        sb, ob = batch_of(src), batch_of(self.work)
        self.ctx.call("uwip_dehaze_histretch", C.byref(sb), C.byref(ob), self.w, DEHAZE_FULL | DEHAZE_GUARD_S,
                      self.letters, 2, 98, 0)

    def stage_aclahe(self):
        wb, vb, ob = batch_of(self.work), batch_of(self.v), batch_of(self.v_out)
        self.ctx.call("uwip_bgr_to_v", C.byref(wb), C.byref(vb))
        # sweep -> parameter choice -> per-frame CLAHE, all queued on the stream (UWIP_ACLAHE_ASYNC: the choice is made on
        # the device and the final CLAHE is launched from the device-side parameters; nothing comes back, the host does not
        # wait -- `params` fetches (BS, CL) when somebody asks).
        # ParametrosACLAHE: the search runs on the 3x3-blurred V (ACLAHE.py:15), the final CLAHE on V itself (main.py:19-20)
        self.ctx.call("uwip_aclahe_auto_ex", C.byref(vb), C.byref(ob), 0, ACLAHE_PREFILTER | ACLAHE_ASYNC, None, None)
        self._params = None
        self.ctx.call("uwip_hsv_replace_v", C.byref(wb), C.byref(ob), C.byref(wb))

    @property
    def params(self):
        """[(BS, CL)] of the most recent aclahe stage (waits for the stream the first time it is read after a stage)"""
        if self._params is None:
            self.ctx.call("uwip_aclahe_last_params", self.h_bs, self.h_cl, self.F)
            self._params = list(zip(self.h_bs, self.h_cl))
        return self._params

    def stage_overlap(self):
        wb = batch_of(self.work)
        if not self.have_prev:
            # first batch: frame 0 is its own key frame (main.cpp:284-297 takes the first frame as key frame)
            pb = batch_of(self.work[0:1])
            self.ctx.call("uwip_overlap_detect", C.byref(pb), self.feats, 0)
            self.have_prev = True
        else:
            # the previous batch's last frame becomes the key frame of this batch's first frame
            self.ctx.call("uwip_features_copy", self.feats, self.F, self.feats, 0)
        self.ctx.call("uwip_overlap_detect", C.byref(wb), self.feats, 1)
        self.ctx.call("uwip_overlap_match", self.feats, self.feats, self.pair_q, self.pair_t, self.F, self.vw, self.vh,
                      self.seed, C.c_void_p(self.ratio.data_ptr()), C.c_void_p(self.info.data_ptr()), None, None, None)

    def stages(self):
        return ["bgdehaze(adaptiveExp_map,w=15)", "histretch(RGB,2/98)", "aclahe(V:blur3+sweep+select+CLAHE,HSV->BGR)",
                "videostrip-overlap(frame vs predecessor: detect+describe+MFMA match+RANSAC+overlapArea)"]

    # ---- host-buffer front end (GpuMat::upload ... download, histretch.cpp:174-175,212-213) ------------
    def host_buffers(self):
        """Page-locked input / output frame buffers [F,H,W,3] owned by the library (freed by ``close``)."""
        shape = (self.F, self.H, self.W, 3)
        a, b = self.ctx.host_alloc(shape), self.ctx.host_alloc(shape)
        self._host_bufs += [a, b]
        return a, b

    def _mark(self, name):
        """timeline instrumentation (tools/host_timeline.py): a timing event on the compute stream, kept with its name"""
        if self.timeline is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(self.stream)
            self.timeline.append((name, self.host.k if self.host else 0, e))

    def _host_state(self):
        """Two source buffers and two result buffers in HBM, so that batch k+1 arrives and batch k leaves while the
        kernels of batch k / k+1 run.  The copies are the copier's (one upload and one download lane per rank, shared by
        all sub-batch pipes when one is passed in); every hand-over is a ticket waited for on the host -- no stream ever
        waits for another one on the device (host-buffer mode)."""
        if self.host is None:
            h = type("HostState", (), {})()
            if self.copier is None:
                self.copier, self._own_copier = Copier(self.dev.index), True
            h.src = [torch.empty((self.F, self.H, self.W, 3), dtype=torch.uint8, device=self.dev) for _ in range(2)]
            h.work = [self.work, torch.empty_like(self.work)]
            h.t_up = [0, 0]              # ticket of the upload into src[i]
            h.t_dn = [0, 0]              # ticket of the download out of work[i]
            h.k = 0
            h.pending = None             # host array whose upload into src[k % 2] has been requested
            self.host = h
        return self.host

    def run_host(self, h_in, h_out, prefetch=None):
        """upload -> the four stages -> download; `h_in` / `h_out` come from ``host_buffers``.  With ``prefetch`` (the
        NEXT batch's input buffer) that batch's upload is requested now and runs under this batch's kernels, so a stream
        of batches never waits for the link.  Returns (upload ticket, download ticket) without waiting for the device:
        ``wait_ticket(t)`` blocks until that copy is complete -- `h_in` may be refilled after the first, `h_out` read
        after the second -- and ``sync()`` drains everything."""
        h = self._host_state()
        k, slot = h.k, h.k % 2
        if h.pending is not h_in:                                    # nobody prefetched this batch
            # src[slot] was last read by batch k-2's dehaze, which precedes everything queued on the stream now
            h.t_up[slot] = self.copier.upload(h.src[slot], h_in, after=self.ctx if k >= 2 else None)
        h.pending = None
        t_in = h.t_up[slot]
        if prefetch is not None:
            # src[1-slot] was last read by batch k-1's dehaze: the upload starts when the stream has finished batch k-1
            h.t_up[1 - slot] = self.copier.upload(h.src[1 - slot], prefetch, after=self.ctx if k >= 1 else None)
            h.pending = prefetch
        self.copier.wait(t_in)                                       # batch k is in HBM
        self.copier.wait(h.t_dn[slot])                               # work[slot] (batch k-2's result) has left
        self.work = h.work[slot]
        self._mark("dehaze0")
        self.stage_dehaze_histretch(h.src[slot])
        self._mark("dehaze1")
        self.stage_aclahe()
        self._mark("aclahe1")
        # the enhanced frames are final here (the overlap stage only reads them): they leave under its kernels
        h.t_dn[slot] = self.copier.download(h_out, self.work, after=self.ctx)
        self.stage_overlap()
        self._mark("overlap1")
        h.k = k + 1
        return t_in, h.t_dn[slot]

    def wait_ticket(self, ticket):
        self.copier.wait(ticket)

    def sync(self):
        """Drain the pipe's stream and, when the host-buffer front end is in use, its outstanding copies."""
        self.ctx.sync()
        if self.host is not None:
            for t in self.host.t_up + self.host.t_dn:
                self.copier.wait(t)

    def run(self, src: torch.Tensor):
        # Nothing in a step waits on the host any more (round 4), so a caller that loops would queue steps without bound
        # and end up spinning inside the runtime once its hardware queue is full: at most two steps are kept in flight,
        # the wait for the third-last one polls its event and sleeps in between.
        self._throttle()
        self.stage_dehaze_histretch(src)
        self.stage_aclahe()
        self.stage_overlap()
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self._inflight.append(ev)
        return self.work, self.ratio

    def _throttle(self, keep: int = 2):
        import time
        while len(self._inflight) >= keep:
            ev = self._inflight.pop(0)
            while not ev.query():
                time.sleep(1e-3)            # two steps are in flight: a millisecond of slack costs nothing
