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
