"""Deterministic synthetic frames: no files are needed
on the GPU box.  seed = 1234 + frame_index; 8UC3 BGR interleaved."""
from __future__ import annotations

import numpy as np


def uw_frame(index: int, rows: int, cols: int, seed0: int = 1234) -> np.ndarray:
    """"uw-like" frame: low-frequency field (6 random 2-D cosines, periods
    80-600 px) + per-channel attenuation (R x0.35, G x0.8, B x1.0; offsets
    +10/+40/+60) + uniform noise +-6, clipped to [0,255]."""
    rng = np.random.default_rng(seed0 + index)
    yy, xx = np.meshgrid(np.arange(rows, dtype=np.float32), np.arange(cols, dtype=np.float32), indexing="ij")
    field = np.zeros((rows, cols), np.float32)
    for _ in range(6):
        period = rng.uniform(80.0, 600.0)
        theta = rng.uniform(0.0, 2 * np.pi)
        phase = rng.uniform(0.0, 2 * np.pi)
        amp = rng.uniform(0.5, 1.0)
        field += np.float32(amp) * np.cos(np.float32(2 * np.pi / period) * (xx * np.float32(np.cos(theta)) + yy * np.float32(np.sin(theta))) + np.float32(phase))
    field = (field - field.min()) / max(float(field.max() - field.min()), 1e-6) * 180.0
    out = np.empty((rows, cols, 3), np.float32)
    out[..., 0] = field * 1.0 + 60.0
    out[..., 1] = field * 0.8 + 40.0
    out[..., 2] = field * 0.35 + 10.0
    out += rng.integers(-6, 7, size=out.shape).astype(np.float32)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def uw_batch(first: int, frames: int, rows: int, cols: int) -> np.ndarray:
    return np.stack([uw_frame(first + i, rows, cols) for i in range(frames)], axis=0)


def adversarial(kind: str, rows: int, cols: int, seed: int = 7) -> np.ndarray:
    """Parity stress inputs: uniform-random bytes, constant plane (degenerate
    stretch), two-level image (max atomics contention), ramp."""
    rng = np.random.default_rng(seed)
    if kind == "random":
        return rng.integers(0, 256, size=(rows, cols, 3), dtype=np.uint8)
    if kind == "constant":
        return np.full((rows, cols, 3), 77, np.uint8)
    if kind == "two_level":
        a = np.where(rng.random((rows, cols, 1)) < 0.5, 10, 240).astype(np.uint8)
        return np.repeat(a, 3, axis=2)
    if kind == "ramp":
        r = (np.arange(cols, dtype=np.int64) * 256 // max(cols, 1)).astype(np.uint8)
        return np.repeat(np.repeat(r[None, :, None], rows, axis=0), 3, axis=2)
    raise ValueError(kind)


def _texture(seed: int, rows: int, cols: int) -> np.ndarray:
    """Corner-rich grey texture in [0,1]: low-frequency field + random rectangles and discs."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(rows, dtype=np.float32), np.arange(cols, dtype=np.float32), indexing="ij")
    t = np.zeros((rows, cols), np.float32)
    for _ in range(4):
        period = rng.uniform(120.0, 500.0)
        theta = rng.uniform(0.0, 2 * np.pi)
        t += np.cos(np.float32(2 * np.pi / period) * (xx * np.float32(np.cos(theta)) + yy * np.float32(np.sin(theta))) + np.float32(rng.uniform(0, 6.28)))
    t = (t - t.min()) / max(float(t.max() - t.min()), 1e-6) * 0.5 + 0.1
    nblobs = max(40, rows * cols // 6000)
    for _ in range(nblobs):
        cy, cx = rng.integers(0, rows), rng.integers(0, cols)
        a = rng.uniform(0.15, 0.45) * (1 if rng.random() < 0.5 else -1)
        if rng.random() < 0.5:
            hh, ww = rng.integers(6, 40), rng.integers(6, 40)
            t[max(cy - hh, 0):cy + hh, max(cx - ww, 0):cx + ww] += np.float32(a)
        else:
            r = rng.integers(5, 28)
            y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, rows), max(cx - r, 0), min(cx + r + 1, cols)
            m = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= r * r
            t[y0:y1, x0:x1][m] += np.float32(a)
    return np.clip(t, 0.0, 1.0)


def uw_stream(first: int, frames: int, rows: int, cols: int, step_frac: float = 0.03, seed0: int = 1234):
    """Consecutive frames of a camera translating over one corner-rich scene: frame i is the
    window of a larger texture at integer offset (dx_i, dy_i) = i * (step_frac*cols, step_frac*cols/3),
    so the true homography between frames i and j is a pure translation.  Underwater colour cast
    and per-frame noise (seed = seed0 + first + i) as in uw_frame."""
    sx = max(1, int(round(step_frac * cols)))
    sy = max(1, sx // 3)
    total = first + frames
    tex = _texture(seed0, rows + sy * (total + 1), cols + sx * (total + 1))
    out = np.empty((frames, rows, cols, 3), np.uint8)
    for i in range(frames):
        k = first + i
        win = tex[k * sy:k * sy + rows, k * sx:k * sx + cols] * 200.0
        rng = np.random.default_rng(seed0 + k)
        f = np.empty((rows, cols, 3), np.float32)
        f[..., 0] = win * 1.0 + 50.0
        f[..., 1] = win * 0.8 + 35.0
        f[..., 2] = win * 0.35 + 10.0
        f += rng.integers(-4, 5, size=f.shape).astype(np.float32)
        out[i] = np.clip(np.rint(f), 0, 255).astype(np.uint8)
    return out


def uw_stream_shift(cols: int, step_frac: float = 0.03):
    sx = max(1, int(round(step_frac * cols)))
    return sx, max(1, sx // 3)


# ---- camera motion with a known homography (translation 2-4 % of width + rotation + scale) ---------------
def _affine(theta_deg: float, scale: float, tx: float, ty: float, cx: float, cy: float) -> np.ndarray:
    """frame pixel (x, y) -> texture pixel: c_tex + scale * R(theta) * ((x, y) - c) + t, as a 3x3 matrix (float64);
    (cx, cy) is the frame centre and c_tex = the same point of the texture window at rest."""
    th = np.deg2rad(theta_deg)
    c, s = np.cos(th) * scale, np.sin(th) * scale
    return np.array([[c, -s, cx - c * cx + s * cy + tx], [s, c, cy - s * cx - c * cy + ty], [0.0, 0.0, 1.0]])


def _sample_bilinear(tex: np.ndarray, X: np.ndarray, Y: np.ndarray) -> np.ndarray:
    x0 = np.floor(X).astype(np.int64); y0 = np.floor(Y).astype(np.int64)
    fx = (X - x0).astype(np.float32); fy = (Y - y0).astype(np.float32)
    x0 = np.clip(x0, 0, tex.shape[1] - 2); y0 = np.clip(y0, 0, tex.shape[0] - 2)
    a, b = tex[y0, x0], tex[y0, x0 + 1]
    c, d = tex[y0 + 1, x0], tex[y0 + 1, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def uw_motion_pair(rows: int, cols: int, theta_deg: float = 0.0, scale: float = 1.0, shift_frac=(0.03, 0.01), seed0: int = 1234,
                   noise: int = 4):
    """A key frame and a current frame of one corner-rich scene whose camera moved by a known similarity: the key frame
    is the texture window at rest, the current frame sees the scene rotated by `theta_deg` about the frame centre,
    zoomed by `scale` (> 1: the camera rose, features shrink) and shifted by `shift_frac` of the frame width.
    Returns (key, cur, H) with H the exact 3x3 homography taking CURRENT-frame pixel coordinates to KEY-frame pixel
    coordinates at full resolution (the direction findHomography(obj, scene) estimates, videostrip.cpp:270)."""
    margin = int(0.75 * max(rows, cols))
    tex = _texture(seed0, rows + 2 * margin, cols + 2 * margin)
    yy, xx = np.meshgrid(np.arange(rows, dtype=np.float64), np.arange(cols, dtype=np.float64), indexing="ij")
    cx, cy = (cols - 1) / 2.0, (rows - 1) / 2.0
    A_key = _affine(0.0, 1.0, 0.0, 0.0, cx, cy)
    A_cur = _affine(theta_deg, scale, shift_frac[0] * cols, shift_frac[1] * cols, cx, cy)
    out = []
    for k, A in enumerate((A_key, A_cur)):
        X = A[0, 0] * xx + A[0, 1] * yy + A[0, 2] + margin
        Y = A[1, 0] * xx + A[1, 1] * yy + A[1, 2] + margin
        win = _sample_bilinear(tex, X, Y) * 200.0
        rng = np.random.default_rng(seed0 + 77 + k)
        f = np.empty((rows, cols, 3), np.float32)
        f[..., 0] = win * 1.0 + 50.0
        f[..., 1] = win * 0.8 + 35.0
        f[..., 2] = win * 0.35 + 10.0
        if noise:
            f += rng.integers(-noise, noise + 1, size=f.shape).astype(np.float32)
        out.append(np.clip(np.rint(f), 0, 255).astype(np.uint8))
    H = np.linalg.inv(A_key) @ A_cur
    return out[0], out[1], H


def to_working_homography(H_full: np.ndarray, cols: int, target_w: int = 640) -> np.ndarray:
    """The same homography between the 640-wide working images: cv::resize maps working pixel x_w to the full-resolution
    position (x_w + 0.5) / f - 0.5 with f = target_w / cols (pixel centres), for both axes."""
    f = target_w / float(cols)
    S = np.array([[f, 0.0, 0.5 * f - 0.5], [0.0, f, 0.5 * f - 0.5], [0.0, 0.0, 1.0]])
    return S @ H_full @ np.linalg.inv(S)


def _motion_state(k: int, cols: int, step_frac: float, seed0: int):
    """Cumulative camera pose of frame k of `uw_stream_motion`: (theta_deg, scale, tx, ty).  Frame j adds a translation of
    step_frac * cols (and a third of it downwards), a yaw drawn uniformly from [-1, 1] degrees and a zoom factor from
    [0.99, 1.01] ("translation 2-4 % of width + rotation <= 1 deg + scale <= 1 %"), each seeded by j alone, so
    any frame can be generated without its predecessors."""
    th, sc = 0.0, 1.0
    for j in range(1, k + 1):
        r = np.random.default_rng(seed0 * 7919 + 104729 + j)
        th += float(r.uniform(-1.0, 1.0))
        sc *= 1.0 + float(r.uniform(-0.01, 0.01))
    sx = step_frac * cols
    return th, sc, k * sx, k * sx / 3.0


def uw_stream_motion(first: int, frames: int, rows: int, cols: int, step_frac: float = 0.03, seed0: int = 1234):
    """SURVEY 8(d)'s synthetic stream in full: consecutive frames of one corner-rich scene related by a KNOWN homography --
    translation, yaw <= 1 degree and zoom <= 1 % per frame (accumulating like a real track) -- sampled bilinearly from a
    larger texture, with the underwater colour cast and per-frame noise of `uw_stream`.  `uw_stream_motion_H` gives the
    exact homography between two frames."""
    total = first + frames
    margin = int(0.2 * max(rows, cols))
    sx = step_frac * cols
    tex = _texture(seed0, rows + int(sx / 3.0 * (total + 1)) + 2 * margin, cols + int(sx * (total + 1)) + 2 * margin)
    yy, xx = np.meshgrid(np.arange(rows, dtype=np.float64), np.arange(cols, dtype=np.float64), indexing="ij")
    cx, cy = (cols - 1) / 2.0, (rows - 1) / 2.0
    out = np.empty((frames, rows, cols, 3), np.uint8)
    for i in range(frames):
        k = first + i
        th, sc, tx, ty = _motion_state(k, cols, step_frac, seed0)
        A = _affine(th, sc, tx, ty, cx, cy)
        X = A[0, 0] * xx + A[0, 1] * yy + A[0, 2] + margin
        Y = A[1, 0] * xx + A[1, 1] * yy + A[1, 2] + margin
        win = _sample_bilinear(tex, X, Y) * 200.0
        rng = np.random.default_rng(seed0 + k)
        f = np.empty((rows, cols, 3), np.float32)
        f[..., 0] = win * 1.0 + 50.0
        f[..., 1] = win * 0.8 + 35.0
        f[..., 2] = win * 0.35 + 10.0
        f += rng.integers(-4, 5, size=f.shape).astype(np.float32)
        out[i] = np.clip(np.rint(f), 0, 255).astype(np.uint8)
    return out


def uw_stream_motion_H(key: int, cur: int, rows: int, cols: int, step_frac: float = 0.03, seed0: int = 1234) -> np.ndarray:
    """Exact 3x3 homography taking pixel coordinates of frame `cur` of `uw_stream_motion` to those of frame `key`."""
    cx, cy = (cols - 1) / 2.0, (rows - 1) / 2.0
    Ak = _affine(*_motion_state(key, cols, step_frac, seed0), cx, cy)
    Ac = _affine(*_motion_state(cur, cols, step_frac, seed0), cx, cy)
    return np.linalg.inv(Ak) @ Ac
