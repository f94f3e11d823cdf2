"""ctypes binding of oracle/liboracle.so for the tests (the oracle is the
checker, never the product).  Builds it with oracle/Makefile when missing."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_numChannel.restype = C.c_int; L.orc_numChannel.argtypes = [C.c_char]
        L.orc_numSpace.restype = C.c_int; L.orc_numSpace.argtypes = [C.c_char]
        L.orc_getHistogram.restype = None
        L.orc_getHistogram.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, f32p]
        L.orc_stretch_lut.restype = None
        L.orc_stretch_lut.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_histretch_bgr.restype = C.c_int
        L.orc_histretch_bgr.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.c_int, C.c_int]
        L.orc_imgChannelStretch.restype = None
        L.orc_imgChannelStretch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int]
        L.orc_bgr_to_v.restype = None
        L.orc_bgr_to_v.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t]
        L.orc_clahe_u8.restype = C.c_int
        L.orc_clahe_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_clahe_tile_geometry.restype = C.c_int
        L.orc_clahe_tile_geometry.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] * 4
        L.orc_aclaheEntropy.restype = C.c_float
        L.orc_aclaheEntropy.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t]
        L.orc_aclahe_sweep.restype = C.c_int
        L.orc_aclahe_sweep.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, f32p]
        L.orc_bgr_to_gray.restype = None
        L.orc_bgr_to_gray.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t]
        L.orc_calcBlur.restype = C.c_float
        L.orc_calcBlur.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t]

        self.KP = np.dtype([("x", "f4"), ("y", "f4"), ("response", "f4"), ("level", "i4"), ("xi", "i4"), ("yi", "i4"),
                            ("co", "f4"), ("si", "f4")])
        L.orc_resize_dims.restype = None
        L.orc_resize_dims.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_resize_gray.restype = None
        L.orc_resize_gray.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, u8p, C.c_void_p]
        L.orc_detect_describe.restype = C.c_int
        L.orc_detect_describe.argtypes = [u8p, C.c_int, C.c_int, C.c_void_p, u8p, C.POINTER(C.c_float)]
        L.orc_detect_describe_ex.restype = C.c_int
        L.orc_detect_describe_ex.argtypes = [u8p, C.c_int, C.c_int, C.c_void_p, u8p, C.POINTER(C.c_float), C.c_int]
        L.orc_scale_space_level.restype = None
        L.orc_scale_space_level.argtypes = [u8p, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p]
        L.orc_match_knn2.restype = None
        L.orc_match_knn2.argtypes = [u8p, C.c_int, u8p, C.c_int, i32p, i32p]
        L.orc_ratio_test.restype = C.c_int
        L.orc_ratio_test.argtypes = [i32p, i32p, C.c_int, C.c_int, i32p, i32p]
        L.orc_find_homography.restype = C.c_int
        L.orc_find_homography.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_double)]
        L.orc_find_homography_ex.restype = C.c_int
        L.orc_find_homography_ex.argtypes = L.orc_find_homography.argtypes + [C.c_int]
        L.orc_overlapArea.restype = C.c_float
        L.orc_overlapArea.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_int32)]
        L.orc_calcOverlap.restype = C.c_float
        L.orc_calcOverlap.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_uint32,
                                      C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        L.orc_calcOverlap_ex.restype = C.c_float
        L.orc_calcOverlap_ex.argtypes = L.orc_calcOverlap.argtypes + [C.c_int]

        L.orc_bgr_to_hsv_px.restype = None
        L.orc_bgr_to_hsv_px.argtypes = [C.c_int] * 3 + [C.POINTER(C.c_int)] * 3
        L.orc_hsv_to_bgr_px.restype = None
        L.orc_hsv_to_bgr_px.argtypes = [C.c_int] * 3 + [u8p]
        L.orc_hsv_replace_v.restype = None
        L.orc_hsv_replace_v.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t, u8p, C.c_size_t]

    # ---- bgdehaze in C (oracle/dehaze_oracle.c): the full-size checker and the CPU baseline ----
    def dehaze(self, img, w=15, full=True, guard_s=False, B=None, taps=()):
        """generate_results() for one uint8 BGR frame.  Returns (out_u8, dict of requested taps):
        taps from {"B", "idx", "traw", "refined", "restored", "final"}."""
        img = np.ascontiguousarray(img)
        M, N = img.shape[:2]
        out = np.zeros_like(img)
        bufs = {"B": np.zeros(3), "idx": np.zeros(2, np.int32), "traw": np.zeros((2, M, N)), "refined": np.zeros((2, M, N)),
                "restored": np.zeros((M, N, 3)), "final": np.zeros((M, N, 3))}
        ptr = lambda k: bufs[k].ctypes.data_as(C.c_void_p) if k in taps else None
        Bi = None if B is None else np.ascontiguousarray(B, np.float64)
        f = self.lib.orc_dehaze_u8
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 6
        rc = f(img.ctypes.data, M, N, img.strides[0], w, (1 if full else 0) | (2 if guard_s else 0),
               None if Bi is None else Bi.ctypes.data, out.ctypes.data, out.strides[0],
               ptr("B"), ptr("idx"), ptr("traw"), ptr("refined"), ptr("restored"), ptr("final"))
        assert rc == 0, rc
        return out, {k: bufs[k] for k in taps}

    def guided_filter(self, guide, p, r=40, eps=1e-3):
        guide, p = np.ascontiguousarray(guide, np.float64), np.ascontiguousarray(p, np.float64)
        q = np.zeros_like(p)
        f = self.lib.orc_guided_filter
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
        rc = f(guide.ctypes.data, p.ctypes.data, p.shape[0], p.shape[1], r, eps, q.ctypes.data)
        assert rc == 0, rc
        return q

    def bgr_to_hsv_px(self, b, g, r):
        v = [C.c_int(0) for _ in range(3)]
        self.lib.orc_bgr_to_hsv_px(b, g, r, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def hsv_to_bgr_px(self, h, s, v):
        out = np.zeros(3, np.uint8)
        self.lib.orc_hsv_to_bgr_px(h, s, v, out)
        return tuple(int(x) for x in out)

    def hsv_replace_v(self, img, vnew):
        img, vnew = np.ascontiguousarray(img), np.ascontiguousarray(vnew)
        out = np.zeros_like(img)
        self.lib.orc_hsv_replace_v(img, img.shape[0], img.shape[1], img.strides[0], vnew, vnew.strides[0], out, out.strides[0])
        return out

    # ---- overlap wrappers ----
    def resize_dims(self, rows, cols, target_w=640):
        a, b = C.c_int(0), C.c_int(0)
        self.lib.orc_resize_dims(rows, cols, target_w, C.byref(a), C.byref(b))
        return a.value, b.value

    def resize_gray(self, img):
        img = np.ascontiguousarray(img)
        oh, ow = self.resize_dims(img.shape[0], img.shape[1])
        g = np.zeros((oh, ow), np.uint8)
        self.lib.orc_resize_gray(img, img.shape[0], img.shape[1], img.strides[0], oh, ow, g, None)
        return g

    def resize_bgr(self, img):
        """cv::resize(frame, res, Size(), f, f), f = 640 / cols: the 8UC3 frame calcBlur receives (main.cpp:311,338)"""
        img = np.ascontiguousarray(img)
        oh, ow = self.resize_dims(img.shape[0], img.shape[1])
        g = np.zeros((oh, ow), np.uint8)
        small = np.zeros((oh, ow, 3), np.uint8)
        self.lib.orc_resize_gray(img, img.shape[0], img.shape[1], img.strides[0], oh, ow, g, small.ctypes.data)
        return small

    def detect_describe(self, gray, upright=False, relative_threshold=False):
        gray = np.ascontiguousarray(gray)
        kps = np.zeros(2048, self.KP)
        desc = np.zeros((2048, 64), np.uint8)
        kc = C.c_float(0)
        n = self.lib.orc_detect_describe_ex(gray, gray.shape[0], gray.shape[1], kps.ctypes.data, desc, C.byref(kc), (1 if upright else 0) | (16 if relative_threshold else 0))
        return kps[:n].copy(), desc[:n].copy(), kc.value

    def scale_space_level(self, gray, level):
        gray = np.ascontiguousarray(gray)
        outs = [np.zeros(gray.shape, np.float32) for _ in range(4)]
        self.lib.orc_scale_space_level(gray, gray.shape[0], gray.shape[1], level, *outs)
        return outs   # Lt, Lx, Ly, Ldet

    def match_knn2(self, dq, dt):
        dq, dt = np.ascontiguousarray(dq), np.ascontiguousarray(dt)
        idx = np.full((max(len(dq), 1), 2), -1, np.int32)
        dist = np.full((max(len(dq), 1), 2), -1, np.int32)
        self.lib.orc_match_knn2(dq.reshape(-1), len(dq), dt.reshape(-1), len(dt), idx.reshape(-1), dist.reshape(-1))
        return idx[:len(dq)], dist[:len(dq)]

    def ratio_test(self, idx, dist, nt):
        nq = len(idx)
        gq, gt = np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.int32)
        n = self.lib.orc_ratio_test(np.ascontiguousarray(idx).reshape(-1), np.ascontiguousarray(dist).reshape(-1), nq, nt, gq, gt)
        return gq[:n], gt[:n]

    def find_homography(self, ox, oy, sx, sy, w, h, seed=1, min_inliers=4):
        H = (C.c_double * 9)()
        a = [np.ascontiguousarray(v, np.float32) for v in (ox, oy, sx, sy)]
        n = self.lib.orc_find_homography_ex(a[0], a[1], a[2], a[3], len(a[0]), w, h, seed, H, min_inliers)
        return n, np.array(list(H)).reshape(3, 3)

    def overlapArea(self, H, vw, vh):
        Hc = (C.c_double * 9)(*np.asarray(H, np.float64).reshape(-1))
        cnt = C.c_int32(0)
        r = self.lib.orc_overlapArea(Hc, vw, vh, C.byref(cnt))
        return float(r), cnt.value

    def calcOverlap(self, key, obj, vw=None, vh=None, seed=1, upright=False, relative_threshold=False, min6=False):
        key, obj = np.ascontiguousarray(key), np.ascontiguousarray(obj)
        vw = key.shape[1] if vw is None else vw
        vh = key.shape[0] if vh is None else vh
        info = (C.c_int32 * 8)()
        H = (C.c_double * 9)()
        r = self.lib.orc_calcOverlap_ex(key, obj, key.shape[0], key.shape[1], key.strides[0], vw, vh, seed, info, H,
                                        (1 if upright else 0) | (16 if relative_threshold else 0) | (8 if min6 else 0))
        return float(r), list(info)[:5], np.array(list(H)).reshape(3, 3)

    # ---- numpy-friendly wrappers ----
    def numChannel(self, c): return self.lib.orc_numChannel(c.encode()[:1])
    def numSpace(self, c): return self.lib.orc_numSpace(c.encode()[:1])

    def getHistogram(self, plane):
        """plane: HxW uint8 view (any strides along rows; pixel stride = last stride)."""
        assert plane.dtype == np.uint8 and plane.ndim == 2
        out = np.zeros(256, np.float32)
        self.lib.orc_getHistogram(plane.ctypes.data, plane.shape[0], plane.shape[1],
                                  plane.strides[0], plane.strides[1], out)
        return out

    def stretch_lut(self, hist, rows, cols, lo, hi):
        lut = np.zeros(256, np.uint8)
        a, b = C.c_int(0), C.c_int(0)
        self.lib.orc_stretch_lut(np.ascontiguousarray(hist, np.float32), rows, cols, lo, hi, lut, C.byref(a), C.byref(b))
        return lut, a.value, b.value

    def histretch(self, img, letters, lo=2, hi=98):
        out = np.ascontiguousarray(img).copy()
        rc = self.lib.orc_histretch_bgr(out, out.shape[0], out.shape[1], out.strides[0], letters.encode(), lo, hi)
        return out, rc

    def cvt_space(self, img, space, to_bgr=False, opencv32=False):
        """cv::cvtColor BGR2{HSV,HLS,Lab,YCrCb} (space 1..4) or back, 8UC3 (oracle/uwip_oracle_color.c); opencv32: Lab -> BGR
        in OpenCV 3.2's float form instead of 3.4.x's integer one"""
        img = np.ascontiguousarray(img)
        out = np.zeros_like(img)
        f = self.lib.orc_cvt_space_ex
        f.restype = C.c_int
        f.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        assert f(img.reshape(-1), img.shape[0], img.shape[1], img.strides[0], out.reshape(-1), out.strides[0], space, 1 if to_bgr else 0,
                 1 if opencv32 else 0) == 0
        return out

    def histretch_ex(self, img, letters, lo=2, hi=98, fixed_order=False, opencv32=False):
        out = np.ascontiguousarray(img).copy()
        f = self.lib.orc_histretch_bgr_ex
        f.restype = C.c_int
        f.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int]
        assert f(out.reshape(-1), out.shape[0], out.shape[1], out.strides[0], letters.encode(), lo, hi, (1 if fixed_order else 0) | (2 if opencv32 else 0)) == 0
        return out

    def imgChannelStretch(self, plane, lo, hi):
        """in place on a 2-D uint8 view"""
        self.lib.orc_imgChannelStretch(plane.ctypes.data, plane.shape[0], plane.shape[1], plane.strides[0], plane.strides[1], lo, hi)

    def bgr_to_v(self, img):
        img = np.ascontiguousarray(img)
        v = np.zeros(img.shape[:2], np.uint8)
        self.lib.orc_bgr_to_v(img, img.shape[0], img.shape[1], img.strides[0], v, v.strides[0])
        return v

    def clahe(self, plane, clip, gx, gy, rule=0, want_luts=False):
        plane = np.ascontiguousarray(plane)
        dst = np.zeros_like(plane)
        luts = np.zeros((gy * gx, 256), np.uint8)
        rc = self.lib.orc_clahe_u8(plane, plane.shape[0], plane.shape[1], plane.strides[0], dst, dst.strides[0],
                                   float(clip), gx, gy, rule, luts.ctypes.data)
        assert rc == 0
        return (dst, luts) if want_luts else dst

    def tile_geometry(self, rows, cols, gx, gy):
        v = [C.c_int(0) for _ in range(4)]
        self.lib.orc_clahe_tile_geometry(rows, cols, gx, gy, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)  # tw, th, padded_cols, padded_rows

    def entropy(self, plane):
        plane = np.ascontiguousarray(plane)
        return float(self.lib.orc_aclaheEntropy(plane, plane.shape[0], plane.shape[1], plane.strides[0]))

    def sweep(self, plane, rule=0):
        plane = np.ascontiguousarray(plane)
        out = np.zeros(5 * 51, np.float32)
        rc = self.lib.orc_aclahe_sweep(plane, plane.shape[0], plane.shape[1], plane.strides[0], rule, out)
        assert rc == 0
        return out.reshape(5, 51)

    def gaussian3(self, plane, rule=0):
        plane = np.ascontiguousarray(plane)
        out = np.zeros_like(plane)
        f = self.lib.orc_gaussian3_u8
        f.restype = None
        f.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t, C.c_int]
        f(plane, plane.shape[0], plane.shape[1], plane.strides[0], out, out.strides[0], rule)
        return out

    def bgr_to_gray(self, img):
        img = np.ascontiguousarray(img)
        g = np.zeros(img.shape[:2], np.uint8)
        self.lib.orc_bgr_to_gray(img, img.shape[0], img.shape[1], img.strides[0], g, g.strides[0])
        return g

    def calcBlur(self, img):
        img = np.ascontiguousarray(img)
        return float(self.lib.orc_calcBlur(img, img.shape[0], img.shape[1], img.strides[0]))


def assert_u8_differs_only_at_rounding_ties(got_u8, ref_float, eps=1e-6, what=""):
    """`got_u8` must equal rint(clip(ref_float * 255)) except where ref_float * 255 sits within `eps` of a half-integer
    (there a last-place difference between two float64 evaluation orders legitimately decides the rounding), and then by
    one level at most.  Returns the number of such pixels."""
    x = np.asarray(ref_float, np.float64) * 255.0
    want = np.clip(np.rint(np.nan_to_num(x, nan=-1.0)), 0, 255).astype(np.int64)
    d = np.asarray(got_u8).astype(np.int64) - want
    bad = d != 0
    if bad.any():
        assert np.abs(d[bad]).max() <= 1, what
        frac = np.abs(x[bad] - np.floor(x[bad]) - 0.5)
        assert frac.max() <= eps, (what, float(frac.max()), int(bad.sum()))
    return int(bad.sum())


_cached = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load():
    global _cached
    if _cached is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith(".c")]
        if (not os.path.exists(LIB)) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
            build()
        _cached = Oracle(C.CDLL(LIB))
    return _cached
