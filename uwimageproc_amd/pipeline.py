"""The per-frame pipe  bgdehaze -> histretch -> aclahe -> videostrip-overlap  on a
batch of frames resident in HBM.  The chain itself -- stage order, the feature-slot
carry between batches, the two-steps-in-flight throttle, the double-buffered host
front end -- lives behind the C ABI (``uwip_pipe_*``, csrc/pipe.cpp; include/uwip.h
"the whole per-frame chain"); this class is its ctypes caller and owns nothing but the
torch tensors it hands in.

Stage definitions (DESIGN.md "the pipe"):
  bgdehaze   main.py:14-20 / adaptiveExp_map (w = 15); as written by default (a 0/0 in S blackens the frame, B-11),
             guard_s=True substitutes S = 1 there (UWIP_DEHAZE_GUARD_S: what bench.py asks for, and says so)
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

from ._native import Context, Copier, PipeConfig, UwipError, batch_of

DEHAZE_FULL, DEHAZE_GUARD_S = 1, 2              # uwip.h
OVERLAP_MIN6 = 8
ACLAHE_PREFILTER, ACLAHE_ASYNC = 1, 4
PIPE_DEHAZE, PIPE_HISTRETCH, PIPE_ACLAHE, PIPE_OVERLAP, PIPE_ALL = 1, 2, 4, 8, 15


class _DevView:
    """Raw device memory described through ``__cuda_array_interface__`` so that torch can wrap it without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2,
                                         "strides": None}


class FramePipe:
    """``uwip_pipe`` with the library's defaults = the reference's rules; the two deviations are opt-in: ``guard_s``
    (UWIP_DEHAZE_GUARD_S) and ``min6`` (UWIP_OVERLAP_MIN6)."""

    def __init__(self, device: int, frames: int, rows: int, cols: int, letters: str = "RGB", w: int = 15,
                 video_size=None, seed: int = 1, copier: "Copier | None" = None, guard_s: bool = False, min6: bool = False):
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        # share torch's current stream so torch events / synchronize cover our kernels
        self.stream = torch.cuda.current_stream(self.dev)
        self.ctx = Context(device, stream=self.stream.cuda_stream)
        self._l = self.ctx._l
        self.copier = copier
        self._host_bufs = []
        self.F, self.H, self.W = frames, rows, cols
        self.seed = seed
        # the reference's globals videoWidth / videoHeight (main.cpp:238-239); as written they are the
        # full-resolution size
        self.vw, self.vh = video_size if video_size else (cols, rows)
        cfg = PipeConfig()
        self._l.uwip_pipe_config_default(C.byref(cfg), frames, rows, cols)
        cfg.letters = letters.encode()
        cfg.w = w
        cfg.dehaze_flags = DEHAZE_FULL | (DEHAZE_GUARD_S if guard_s else 0)
        cfg.match_flags = OVERLAP_MIN6 if min6 else 0
        cfg.videoWidth, cfg.videoHeight = self.vw, self.vh
        cfg.seed = seed
        # the staging area of the host-buffer form is torch's, so that the results can be looked at as tensors
        self._staging = None
        self._cfg = cfg
        self._p = None
        self.work = torch.empty((frames, rows, cols, 3), dtype=torch.uint8, device=self.dev)
        self.ratio = torch.empty((frames,), dtype=torch.float32, device=self.dev)
        self.info = torch.zeros((frames, 8), dtype=torch.int32, device=self.dev)
        self._own = (self.work, self.ratio, self.info)
        self.h_bs = (C.c_int32 * frames)()
        self.h_cl = (C.c_int32 * frames)()
        self._params = None
        self._k = 0
        self._create()

    def _create(self):
        h = C.c_void_p()
        self.ctx.call("uwip_pipe_create", C.byref(self._cfg), self.copier._h if self.copier is not None else None, C.byref(h))
        self._p = h

    def _call(self, name, *args):
        rc = getattr(self._l, name)(self._p, *args)
        if rc:
            raise UwipError(rc, self._l.uwip_pipe_last_error(self._p).decode("utf-8", "replace"))

    def close(self):
        if getattr(self, "_p", None):
            self._l.uwip_pipe_destroy(self._p)
            self._p = None
        for a in getattr(self, "_host_bufs", []):
            if a.ctypes.data in getattr(self.ctx, "_pinned", {}):
                self.ctx.host_free(a)
        self._host_bufs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stages (uwip_pipe_stages: the chain tapped between its stages) ------------------------------
    def _stages(self, mask, src=None):
        sb = batch_of(src) if src is not None else None
        wb = batch_of(self.work)
        self._call("uwip_pipe_stages", mask, C.byref(sb) if sb is not None else None, C.byref(wb),
                   C.c_void_p(self.ratio.data_ptr()), C.c_void_p(self.info.data_ptr()))
        if mask & PIPE_ACLAHE:
            self._params = None

    def stage_dehaze(self, src: torch.Tensor):
        self._stages(PIPE_DEHAZE, src)

    def stage_histretch(self):
        self._stages(PIPE_HISTRETCH)

    def stage_dehaze_histretch(self, src: torch.Tensor):
        """The two stages as one chained call: the dehaze writer hands the stretch its histogram."""
        self._stages(PIPE_DEHAZE | PIPE_HISTRETCH, src)

    def stage_aclahe(self):
        self._stages(PIPE_ACLAHE)

    def stage_overlap(self):
        self._stages(PIPE_OVERLAP)

    @property
    def params(self):
        """[(BS, CL)] of the most recent aclahe stage (waits for the stream the first time it is read after a stage)"""
        if self._params is None:
            self._call("uwip_pipe_last_params", self.h_bs, self.h_cl)
            self._params = list(zip(self.h_bs, self.h_cl))
        return self._params

    @property
    def have_prev(self):
        raise AttributeError("write-only: assign False to forget the carried key frame")

    @have_prev.setter
    def have_prev(self, value):
        assert not value
        self._call("uwip_pipe_reset")

    @property
    def v(self):
        """The V planes the aclahe stage left in the pipe ([F, H, W] uint8, device): a view, valid until the next step."""
        p = C.c_void_p()
        self._call("uwip_pipe_device_results", C.byref(p), None, None, None)
        try:
            return torch.as_tensor(_DevView(p.value, (self.F, self.H, self.W), "|u1"), device=self.dev)
        except Exception:
            import numpy as np                       # a torch without __cuda_array_interface__: through the host
            h = np.empty((self.F, self.H, self.W), np.uint8)
            self.ctx.call("uwip_memcpy_d2h", C.c_void_p(h.ctypes.data), p, h.nbytes)
            return torch.from_numpy(h).to(self.dev)

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

    def _host_state(self):
        if self._staging is None:
            # uwip_pipe_config.d_staging must be known at creation: the pipe is made anew (nothing has been carried yet)
            assert self._k == 0
            n = self._l.uwip_pipe_staging_bytes(C.byref(self._cfg))
            self._staging = torch.empty((n,), dtype=torch.uint8, device=self.dev)
            self._cfg.d_staging = self._staging.data_ptr()
            self._l.uwip_pipe_destroy(self._p)
            self._create()
            fb = self.F * self.H * self.W * 3
            al = lambda x: (x + 255) & ~255
            st = self._staging
            self._h_work = [st[(2 + s) * fb:(3 + s) * fb].view(self.F, self.H, self.W, 3) for s in range(2)]
            r0 = al(4 * fb)
            self._h_ratio = [st[r0 + 4 * self.F * s:r0 + 4 * self.F * (s + 1)].view(torch.float32) for s in range(2)]
            i0 = r0 + al(8 * self.F)
            self._h_info = [st[i0 + 32 * self.F * s:i0 + 32 * self.F * (s + 1)].view(torch.int32).view(self.F, 8) for s in range(2)]

    def run_host(self, h_in, h_out, prefetch=None, h_ratio=None):
        """upload -> the four stages -> download; `h_in` / `h_out` come from ``host_buffers``.  With ``prefetch`` (the
        NEXT batch's input buffer) that batch's upload is requested now and runs under this batch's kernels, so a stream
        of batches never waits for the link.  Returns (upload ticket, download ticket) without waiting for the device:
        ``wait_ticket(t)`` blocks until that copy is complete -- `h_in` may be refilled after the first, `h_out` read
        after the second -- and ``sync()`` drains everything.  (uwip_pipe_step_host.)"""
        self._host_state()
        t = (C.c_uint64 * 3)()
        self._call("uwip_pipe_step_host", C.c_void_p(h_in.ctypes.data), C.c_void_p(h_out.ctypes.data),
                   C.c_void_p(h_ratio.ctypes.data) if h_ratio is not None else None,
                   C.c_void_p(prefetch.ctypes.data) if prefetch is not None else None, t)
        slot = self._k % 2
        self.work, self.ratio, self.info = self._h_work[slot], self._h_ratio[slot], self._h_info[slot]
        self._k += 1
        self._params = None
        return t[0], t[1]

    def wait_ticket(self, ticket):
        self._call("uwip_pipe_wait", C.c_uint64(ticket))

    def sync(self):
        """Drain the pipe's stream and, when the host-buffer front end is in use, its outstanding copies."""
        self._call("uwip_pipe_sync")

    def run(self, src: torch.Tensor):
        """One step on frames resident in HBM (uwip_pipe_step: queued without a host wait, at most two steps in flight)."""
        self.work, self.ratio, self.info = self._own
        sb, wb = batch_of(src), batch_of(self.work)
        self._call("uwip_pipe_step", C.byref(sb), C.byref(wb), C.c_void_p(self.ratio.data_ptr()), C.c_void_p(self.info.data_ptr()))
        self._params = None
        return self.work, self.ratio
