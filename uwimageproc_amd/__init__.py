"""uwimageproc_amd -- MI355X-native (gfx950) hot path of uwimageproc.

bgdehaze -> histretch -> aclahe -> videostrip-overlap, as hand-written HIP
kernels behind the C ABI in include/uwip.h.  The Python modules here mirror
the reference's host-side interfaces (same names, argument meaning and error
behaviour) on top of that ABI; they hold no compute of their own and there is
no CPU fallback.
"""
from ._native import BatchU8, Context, Copier, PipeConfig, UwipError, batch_of, device_count, lib  # noqa: F401

__all__ = ["BatchU8", "Context", "Copier", "PipeConfig", "UwipError", "batch_of", "device_count", "lib"]
