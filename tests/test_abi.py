"""CPU: the C-ABI library loads here (no GPU) and exports every symbol that
include/uwip.h declares; compute entry points fail loudly without a device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names += re.findall(r"\b(uwip_[A-Za-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_something():
    d = _declared()
    assert "uwip_histretch" in d and "uwip_clahe" in d and len(d) >= 20


def test_library_exports_every_declared_symbol():
    import uwimageproc_amd._native as nat
    assert os.path.exists(nat.LIB_PATH), "build libuwip.so first (python -c 'import __graft_entry__ as g; g.build()')"
    l = C.CDLL(nat.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(l, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_binding_covers_every_declared_symbol():
    import uwimageproc_amd._native as nat
    assert sorted(nat.SIGNATURES) == _declared()
    nat.lib()   # resolves all of them


def test_host_only_entry_points():
    import uwimageproc_amd._native as nat
    l = nat.lib()
    assert l.uwip_version().startswith(b"uwip")
    assert [l.uwip_numChannel(c.encode()) for c in "RGBx"] == [0, 1, 2, -1]
    assert [l.uwip_numSpace(c.encode()) for c in "RHhLYr"] == [0, 1, 2, 3, 4, -1]


def test_no_cpu_fallback_without_device():
    import torch
    import uwimageproc_amd as uw
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(uw.UwipError):
        uw.Context(0)
    with pytest.raises(uw.UwipError):
        uw.Copier(0)                     # the copy engine needs a device as well: no host-only fallback


def test_copier_argument_checks_without_device():
    """uwip_copier_* reject null handles / null out-pointers with UWIP_ERR_INVALID instead of touching them."""
    import uwimageproc_amd._native as nat
    l = nat.lib()
    t = C.c_uint64(7)
    d = C.c_int(5)
    assert l.uwip_copier_create(0, None) == nat.UWIP_ERR_INVALID
    assert l.uwip_copier_upload(None, None, None, None, 16, C.byref(t)) == nat.UWIP_ERR_INVALID
    assert l.uwip_copier_download(None, None, None, None, 16, C.byref(t)) == nat.UWIP_ERR_INVALID
    assert l.uwip_copier_wait(None, 0) == nat.UWIP_ERR_INVALID
    assert l.uwip_copier_query(None, 0, C.byref(d)) == nat.UWIP_ERR_INVALID
    assert l.uwip_copier_destroy(None) == nat.UWIP_OK
    assert l.uwip_copier_last_error(None) == b"null copier"
