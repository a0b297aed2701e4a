"""Deterministic synthetic inputs ("speckle-warp v1", SURVEY.md section 8d / BASELINE.md section 3).

The reference's only DICOM is a missing blob, so every benchmark/parity input is generated here:
  base = gaussian_filter(standard_normal((H+64, W+64)), sigma=2.5) -> min-max to [0,255]
  truth flow u = tx + s(x-cx), v = ty + s(y-cy), tx,ty ~ U(-2,2), s ~ U(-0.01,0.01)
  I0 = crop(base), I1 = crop(base resampled at (x-u, y-v), cubic); both rounded to uint8.
"""
import numpy as np


def speckle_pair(seed, H=512, W=512, sigma=2.5):
    """Return (I0 uint8[H,W], I1 uint8[H,W], truth float32[H,W,2]) with I1(x+u, y+v) ~= I0(x, y)."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    pad = 32
    base = ndimage.gaussian_filter(rng.standard_normal((H + 2 * pad, W + 2 * pad)), sigma)
    base = (base - base.min()) / (base.max() - base.min()) * 255.0
    tx, ty = rng.uniform(-2, 2, 2)
    s = rng.uniform(-0.01, 0.01)
    yy, xx = np.mgrid[0:H + 2 * pad, 0:W + 2 * pad].astype(np.float64)
    cx, cy = pad + (W - 1) / 2.0, pad + (H - 1) / 2.0
    u = tx + s * (xx - cx)
    v = ty + s * (yy - cy)
    warped = ndimage.map_coordinates(base, [yy - v, xx - u], order=3, mode="nearest")
    crop = (slice(pad, pad + H), slice(pad, pad + W))
    I0 = np.rint(base[crop]).clip(0, 255).astype(np.uint8)
    I1 = np.rint(warped[crop]).clip(0, 255).astype(np.uint8)
    truth = np.stack([u[crop], v[crop]], -1).astype(np.float32)
    return I0, I1, truth


def speckle_pairs(seeds, H=512, W=512):
    """Batch of independent pairs: (I0s uint8[B,H,W], I1s uint8[B,H,W])."""
    a = [speckle_pair(s, H, W)[:2] for s in seeds]
    return np.stack([p[0] for p in a]), np.stack([p[1] for p in a])


def speckle_sequence(seed, N, H=512, W=512, sigma=2.5):
    """A study-like stack: N uint8 frames of one texture under a slowly varying affine motion."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    pad = 48
    base = ndimage.gaussian_filter(rng.standard_normal((H + 2 * pad, W + 2 * pad)), sigma)
    base = (base - base.min()) / (base.max() - base.min()) * 255.0
    yy, xx = np.mgrid[0:H + 2 * pad, 0:W + 2 * pad].astype(np.float64)
    cx, cy = pad + (W - 1) / 2.0, pad + (H - 1) / 2.0
    phase = rng.uniform(0, 2 * np.pi)
    frames = np.empty((N, H, W), np.uint8)
    crop = (slice(pad, pad + H), slice(pad, pad + W))
    for i in range(N):
        t = 2 * np.pi * i / max(N, 2) + phase
        tx, ty, s = 6 * np.sin(t), 4 * np.cos(t), 0.02 * np.sin(t)
        u = tx + s * (xx - cx)
        v = ty + s * (yy - cy)
        w = ndimage.map_coordinates(base, [yy - v, xx - u], order=3, mode="nearest")
        frames[i] = np.rint(w[crop]).clip(0, 255).astype(np.uint8)
    return frames
