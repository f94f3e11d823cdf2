"""ctypes binding of libuwip.so (the C ABI declared in include/uwip.h).

This is plumbing only: it loads the in-tree shared library built by
``uwimageproc_amd/csrc/Makefile`` and exposes the C entry points.  There is no
Python/CPU fallback -- if the library is missing, or no HIP device is present
when a context is created, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuwip.so")

UWIP_OK = 0
UWIP_ERR_INVALID = 1
UWIP_ERR_HIP = 2
UWIP_ERR_UNSUPPORTED = 3
UWIP_ERR_NOMEM = 4


class UwipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"uwip error {code}: {msg}")
        self.code = code


class BatchU8(C.Structure):
    """Mirror of ``uwip_batch_u8`` (cv::Mat fields + batch extent)."""

    _fields_ = [
        ("data", C.c_void_p),
        ("step", C.c_size_t),
        ("frame_stride", C.c_size_t),
        ("rows", C.c_int32),
        ("cols", C.c_int32),
        ("channels", C.c_int32),
        ("frames", C.c_int32),
    ]


class PipeConfig(C.Structure):
    """Mirror of ``uwip_pipe_config``."""

    _fields_ = [
        ("frames", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
        ("letters", C.c_char * 16),
        ("lo", C.c_int32), ("hi", C.c_int32),
        ("w", C.c_int32),
        ("dehaze_flags", C.c_uint32), ("histretch_flags", C.c_uint32),
        ("residual_rule", C.c_int32),
        ("aclahe_flags", C.c_uint32), ("detect_flags", C.c_uint32), ("match_flags", C.c_uint32),
        ("videoWidth", C.c_int32), ("videoHeight", C.c_int32),
        ("seed", C.c_uint32),
        ("max_in_flight", C.c_int32),
        ("d_staging", C.c_void_p),
    ]


_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); every symbol include/uwip.h declares
_P = C.c_void_p
_B = C.POINTER(BatchU8)
SIGNATURES = {
    "uwip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "uwip_ctx_create": (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    "uwip_ctx_create_ex": (C.c_int, [C.c_int, _P, C.c_uint, C.POINTER(_P)]),
    "uwip_ctx_destroy": (C.c_int, [_P]),
    "uwip_last_error": (C.c_char_p, [_P]),
    "uwip_version": (C.c_char_p, []),
    "uwip_sync": (C.c_int, [_P]),
    "uwip_malloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "uwip_free": (C.c_int, [_P, _P]),
    "uwip_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "uwip_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "uwip_host_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "uwip_host_free": (C.c_int, [_P, _P]),
    "uwip_memcpy_h2d_async": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "uwip_memcpy_d2h_async": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "uwip_copier_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "uwip_copier_destroy": (C.c_int, [_P]),
    "uwip_copier_upload": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.POINTER(C.c_uint64)]),
    "uwip_copier_download": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.POINTER(C.c_uint64)]),
    "uwip_copier_wait": (C.c_int, [_P, C.c_uint64]),
    "uwip_copier_query": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_int)]),
    "uwip_copier_last_error": (C.c_char_p, [_P]),
    "uwip_prof_enable": (C.c_int, [_P, C.c_int]),
    "uwip_prof_reset": (C.c_int, [_P]),
    "uwip_prof_count": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "uwip_prof_get": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "uwip_numChannel": (C.c_int, [C.c_char]),
    "uwip_numSpace": (C.c_int, [C.c_char]),
    "uwip_getHistogram": (C.c_int, [_P, _B, _P]),
    "uwip_stretch_lut": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "uwip_apply_lut": (C.c_int, [_P, _B, _P]),
    "uwip_imgChannelStretch": (C.c_int, [_P, _B, C.c_int, C.c_int, C.c_int]),
    "uwip_histretch": (C.c_int, [_P, _B, C.c_char_p, C.c_int, C.c_int]),
    "uwip_histretch_ex": (C.c_int, [_P, _B, C.c_char_p, C.c_int, C.c_int, C.c_uint]),
    "uwip_cvtColor": (C.c_int, [_P, _B, _B, C.c_int, C.c_int]),
    "uwip_cvtColor_ex": (C.c_int, [_P, _B, _B, C.c_int, C.c_int, C.c_int]),
    "uwip_bgr_to_v": (C.c_int, [_P, _B, _B]),
    "uwip_clahe": (C.c_int, [_P, _B, _B, C.c_double, C.c_int, C.c_int, C.c_int]),
    "uwip_clahe_per_frame": (C.c_int, [_P, _B, _B, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int]),
    "uwip_clahe_luts": (C.c_int, [_P, _B, C.c_double, C.c_int, C.c_int, C.c_int, _P]),
    "uwip_entropy": (C.c_int, [_P, _B, _P]),
    "uwip_aclahe_sweep": (C.c_int, [_P, _B, C.c_int, _P]),
    "uwip_aclahe_sweep_hist": (C.c_int, [_P, _B, C.c_int, _P, _P]),
    "uwip_aclahe_knee": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "uwip_aclahe_last_params": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]),
    "uwip_aclahe_select_device": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "uwip_host_pool_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "uwip_aclahe_select": (C.c_int, [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "uwip_aclahe_auto": (C.c_int, [_P, _B, _B, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "uwip_aclahe_auto_ex": (C.c_int, [_P, _B, _B, C.c_int, C.c_uint, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "uwip_GaussianBlur3": (C.c_int, [_P, _B, _B, C.c_int]),
    "uwip_hsv_replace_v": (C.c_int, [_P, _B, _B, _B]),
    "uwip_dehaze_background_light": (C.c_int, [_P, _B, C.c_int, _P, _P]),
    "uwip_dehaze_transmission": (C.c_int, [_P, _B, _P, _P]),
    "uwip_guided_filter": (C.c_int, [_P, _B, _P, C.c_int, C.c_double, _P]),
    "uwip_dehaze": (C.c_int, [_P, _B, _B, C.c_int, C.c_int, _P, _P, _P]),
    "uwip_dehaze_histretch": (C.c_int, [_P, _B, _B, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_uint]),
    "uwip_features_create": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "uwip_features_destroy": (C.c_int, [_P]),
    "uwip_features_copy": (C.c_int, [_P, _P, C.c_int, _P, C.c_int]),
    "uwip_overlap_working_size": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "uwip_resize_bgr": (C.c_int, [_P, _B, _B]),
    "uwip_overlap_detect": (C.c_int, [_P, _B, _P, C.c_int]),
    "uwip_overlap_detect_ex": (C.c_int, [_P, _B, _P, C.c_int, C.c_uint]),
    "uwip_features_download": (C.c_int, [_P, _P, C.c_int, _P, _P, C.POINTER(C.c_int32)]),
    "uwip_features_upload": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int32]),
    "uwip_overlap_debug_level": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "uwip_overlap_match": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int,
                                     C.c_uint32, _P, _P, _P, _P, _P]),
    "uwip_overlap_match_ex": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int,
                                        C.c_uint32, C.c_uint, _P, _P, _P, _P, _P]),
    "uwip_overlapArea": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "uwip_calcBlur": (C.c_int, [_P, _B, _P]),
    "uwip_pipe_config_default": (C.c_int, [C.POINTER(PipeConfig), C.c_int, C.c_int, C.c_int]),
    "uwip_pipe_staging_bytes": (C.c_size_t, [C.POINTER(PipeConfig)]),
    "uwip_pipe_create": (C.c_int, [_P, C.POINTER(PipeConfig), _P, C.POINTER(_P)]),
    "uwip_pipe_destroy": (C.c_int, [_P]),
    "uwip_pipe_last_error": (C.c_char_p, [_P]),
    "uwip_pipe_step": (C.c_int, [_P, _B, _B, _P, _P]),
    "uwip_pipe_stages": (C.c_int, [_P, C.c_uint, _B, _B, _P, _P]),
    "uwip_pipe_step_host": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(C.c_uint64)]),
    "uwip_pipe_wait": (C.c_int, [_P, C.c_uint64]),
    "uwip_pipe_sync": (C.c_int, [_P]),
    "uwip_pipe_reset": (C.c_int, [_P]),
    "uwip_pipe_last_params": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "uwip_pipe_device_results": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
}


def lib() -> C.CDLL:
    """Load libuwip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C uwimageproc_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback"
            )
        # torch first: it brings its own HIP runtime (torch/lib/libamdhip64.so); loading libuwip.so before it would map
        # /opt/rocm's copy as well, and with two runtimes in one process the second one finds no device
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def device_count() -> int:
    n = C.c_int(0)
    lib().uwip_device_count(C.byref(n))
    return n.value


class Context:
    """Owns a ``uwip_ctx``.  ``stream`` = None: the context creates its own stream; an int: that raw
    hipStream_t handle is used as it is -- including 0, the device's default stream (which is what
    ``torch.cuda.current_stream().cuda_stream`` is until a side stream is made current)."""

    UWIP_CTX_STREAM_GIVEN = 1

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._l = lib()
        h = _P()
        if stream is None:
            rc = self._l.uwip_ctx_create(int(device), None, C.byref(h))
        else:
            rc = self._l.uwip_ctx_create_ex(int(device), _P(int(stream)), self.UWIP_CTX_STREAM_GIVEN, C.byref(h))
        if rc != UWIP_OK:
            raise UwipError(rc, "uwip_ctx_create failed (no HIP device? there is no CPU fallback)")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._l.uwip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc != UWIP_OK:
            raise UwipError(rc, self._l.uwip_last_error(self._h).decode("utf-8", "replace"))

    def call(self, name: str, *args):
        self.check(getattr(self._l, name)(self._h, *args))

    def sync(self):
        self.call("uwip_sync")

    # ---- page-locked host buffers + stream-ordered transfers (GpuMat::upload / download with a stream) ----
    def host_alloc(self, shape, dtype="uint8"):
        """A numpy array over page-locked host memory owned by the library (freed by ``host_free``)."""
        import numpy as np

        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = _P()
        self.call("uwip_host_alloc", n, C.byref(p))
        buf = (C.c_uint8 * n).from_address(p.value)
        a = np.frombuffer(buf, dtype=dt).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p
        return a

    def host_free(self, a):
        p = self._pinned.pop(a.ctypes.data)
        self.call("uwip_host_free", p)

    def h2d_async(self, d_tensor, h_array):
        assert d_tensor.is_contiguous() and h_array.flags.c_contiguous
        assert d_tensor.numel() * d_tensor.element_size() == h_array.nbytes
        self.call("uwip_memcpy_h2d_async", _P(d_tensor.data_ptr()), _P(h_array.ctypes.data), h_array.nbytes)

    def d2h_async(self, h_array, d_tensor):
        assert d_tensor.is_contiguous() and h_array.flags.c_contiguous
        assert d_tensor.numel() * d_tensor.element_size() == h_array.nbytes
        self.call("uwip_memcpy_d2h_async", _P(h_array.ctypes.data), _P(d_tensor.data_ptr()), h_array.nbytes)

    # ---- profiling ----
    def prof_enable(self, on: bool = True):
        self.call("uwip_prof_enable", 1 if on else 0)

    def prof_reset(self):
        self.call("uwip_prof_reset")

    def prof_results(self) -> dict:
        n = C.c_int(0)
        self.call("uwip_prof_count", C.byref(n))
        out = {}
        for i in range(n.value):
            name = C.create_string_buffer(128)
            ms = C.c_double(0)
            cnt = C.c_uint64(0)
            self.call("uwip_prof_get", i, name, 128, C.byref(ms), C.byref(cnt))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out


class Copier:
    """Owns a ``uwip_copier``: the upload lane and the download lane of one device (include/uwip.h).  ``after`` is a
    Context: the copy starts once everything queued on its stream at the time of the call has finished."""

    def __init__(self, device: int = 0):
        self._l = lib()
        h = _P()
        rc = self._l.uwip_copier_create(int(device), C.byref(h))
        if rc != UWIP_OK:
            raise UwipError(rc, "uwip_copier_create failed (no HIP device? there is no CPU fallback)")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._l.uwip_copier_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != UWIP_OK:
            raise UwipError(rc, self._l.uwip_copier_last_error(self._h).decode("utf-8", "replace"))

    def upload(self, d_tensor, h_array, after: "Context | None" = None) -> int:
        assert d_tensor.is_contiguous() and h_array.flags.c_contiguous
        assert d_tensor.numel() * d_tensor.element_size() == h_array.nbytes
        t = C.c_uint64(0)
        self._check(self._l.uwip_copier_upload(self._h, after._h if after is not None else None, _P(d_tensor.data_ptr()),
                                               _P(h_array.ctypes.data), h_array.nbytes, C.byref(t)))
        return t.value

    def download(self, h_array, d_tensor, after: "Context | None" = None) -> int:
        assert d_tensor.is_contiguous() and h_array.flags.c_contiguous
        assert d_tensor.numel() * d_tensor.element_size() == h_array.nbytes
        t = C.c_uint64(0)
        self._check(self._l.uwip_copier_download(self._h, after._h if after is not None else None, _P(h_array.ctypes.data),
                                                 _P(d_tensor.data_ptr()), h_array.nbytes, C.byref(t)))
        return t.value

    def wait(self, ticket: int):
        self._check(self._l.uwip_copier_wait(self._h, C.c_uint64(ticket)))

    def done(self, ticket: int) -> bool:
        d = C.c_int(0)
        self._check(self._l.uwip_copier_query(self._h, C.c_uint64(ticket), C.byref(d)))
        return bool(d.value)


def batch_of(t) -> BatchU8:
    """Describe a torch uint8 CUDA tensor [F,H,W,C], [F,H,W], [H,W,C] or [H,W]
    (row-major, unit stride on the last axes) as a ``uwip_batch_u8``."""
    import torch

    assert t.dtype == torch.uint8, "uint8 tensor expected"
    assert t.is_cuda, "device tensor expected (no CPU path)"
    shape, st = list(t.shape), list(t.stride())
    if t.dim() == 2:
        shape, st = [1] + shape + [1], [0] + st + [1]
    elif t.dim() == 3:
        if shape[-1] in (1, 3) and st[-1] == 1 and st[-2] == shape[-1]:
            shape, st = [1] + shape, [0] + st
        else:
            shape, st = shape + [1], st + [1]
    assert len(shape) == 4
    F, H, W, Cn = shape
    assert Cn in (1, 3)
    # strides of size-1 axes carry no information
    if Cn == 1:
        st[3] = 1
    if W == 1:
        st[2] = Cn
    if H * W * F > 0:
        assert st[3] == 1 and st[2] == Cn, "pixels must be packed"
    b = BatchU8()
    b.data = t.data_ptr()
    b.step = st[1] if H > 1 else W * Cn
    b.frame_stride = st[0] if F > 1 else b.step * H
    b.rows, b.cols, b.channels, b.frames = H, W, Cn, F
    return b
