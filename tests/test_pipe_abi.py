"""CPU: the whole-chain entry (uwip_pipe_*, include/uwip.h) -- its configuration defaults are the REFERENCE's rules, the
staging layout is what the header says, and null / malformed arguments are refused without touching a device."""
import ctypes as C

import uwimageproc_amd._native as nat


def _cfg(frames=4, rows=270, cols=480):
    l = nat.lib()
    c = nat.PipeConfig()
    assert l.uwip_pipe_config_default(C.byref(c), frames, rows, cols) == nat.UWIP_OK
    return l, c


def test_defaults_are_the_references_rules():
    l, c = _cfg()
    assert (c.frames, c.rows, c.cols) == (4, 270, 480)
    assert c.letters == b"RGB" and (c.lo, c.hi) == (2, 98)              # histretch.cpp:154,236,247
    assert c.w == 15                                                     # bgdehaze/main.py:28-29
    assert c.dehaze_flags == 1                                           # UWIP_DEHAZE_FULL only: S unguarded (BGDehaze.py:83)
    assert c.match_flags == 0 and c.detect_flags == 0                    # >= 4 good matches (videostrip.cpp:252-272)
    assert c.aclahe_flags == 1 | 4 and c.residual_rule == 0              # ParametrosACLAHE prefilter, no host wait; OpenCV 3.4.x
    assert (c.videoWidth, c.videoHeight) == (0, 0) and c.seed == 1 and c.max_in_flight == 2 and not c.d_staging
    assert l.uwip_pipe_config_default(None, 1, 1, 1) == nat.UWIP_ERR_INVALID


def test_staging_layout():
    l, c = _cfg(frames=3, rows=100, cols=200)
    fb = 3 * 100 * 200 * 3
    al = lambda x: (x + 255) & ~255
    assert l.uwip_pipe_staging_bytes(C.byref(c)) == al(4 * fb) + al(2 * 4 * 3) + al(2 * 8 * 4 * 3)
    c.frames = 0
    assert l.uwip_pipe_staging_bytes(C.byref(c)) == 0                    # a malformed configuration has no staging area
    assert l.uwip_pipe_staging_bytes(None) == 0


def test_null_handles_are_refused():
    l, c = _cfg()
    h = C.c_void_p()
    t = (C.c_uint64 * 3)()
    assert l.uwip_pipe_create(None, C.byref(c), None, C.byref(h)) == nat.UWIP_ERR_INVALID and not h.value
    assert l.uwip_pipe_create(None, C.byref(c), None, None) == nat.UWIP_ERR_INVALID
    assert l.uwip_pipe_step(None, None, None, None, None) == nat.UWIP_ERR_INVALID
    assert l.uwip_pipe_stages(None, 15, None, None, None, None) == nat.UWIP_ERR_INVALID
    assert l.uwip_pipe_step_host(None, None, None, None, None, t) == nat.UWIP_ERR_INVALID
    assert l.uwip_pipe_wait(None, 1) == nat.UWIP_ERR_INVALID
    assert l.uwip_pipe_sync(None) == nat.UWIP_ERR_INVALID
    assert l.uwip_pipe_reset(None) == nat.UWIP_ERR_INVALID
    assert l.uwip_pipe_last_params(None, None, None) == nat.UWIP_ERR_INVALID
    assert l.uwip_pipe_device_results(None, None, None, None, None) == nat.UWIP_ERR_INVALID
    assert l.uwip_pipe_destroy(None) == nat.UWIP_OK
    assert l.uwip_pipe_last_error(None) == b"null pipe"
