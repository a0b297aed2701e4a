#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity for the no_saliency=False preprocessing (tf_saliency_frames): random sizes, frame counts, channel
counts and image statistics (noise, smooth blobs, saturated and flat frames, text-like overlays), maps compared byte for byte.
usage: python tools/fuzz_saliency.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def draw(rng, N, H, W, ch):
    kind = int(rng.integers(0, 6))
    if kind == 0:
        a = rng.integers(0, 256, (N, H, W, ch), dtype=np.uint8)
    elif kind == 1:                                       # smooth blobs
        yy, xx = np.mgrid[0:H, 0:W]
        a = np.zeros((N, H, W, ch), np.float64)
        for _ in range(int(rng.integers(1, 6))):
            cy, cx, s = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(2, max(3, min(H, W) / 2))
            a += (rng.uniform(50, 255) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)))[None, :, :, None]
        a = np.clip(a, 0, 255).astype(np.uint8)
    elif kind == 2:                                       # bright: the float integral image passes 2^24 for larger frames
        a = (255 - rng.integers(0, 24, (N, H, W, ch))).astype(np.uint8)
    elif kind == 3:                                       # mostly flat with a few outliers; one frame entirely flat
        a = np.full((N, H, W, ch), int(rng.integers(0, 256)), np.uint8)
        for _ in range(int(rng.integers(0, 5))):
            a[int(rng.integers(0, N)), int(rng.integers(0, H)), int(rng.integers(0, W))] = int(rng.integers(0, 256))
    elif kind == 4:                                       # echo-like sector: speckle inside a wedge, black outside, a coloured overlay
        yy, xx = np.mgrid[0:H, 0:W]
        wedge = np.abs(xx - W / 2) < (yy + 1) * 0.6
        a = (rng.gamma(2.0, 30.0, (N, H, W, 1)) * wedge[None, :, :, None]).clip(0, 255).astype(np.uint8).repeat(ch, axis=3)
        if ch == 3 and H > 8 and W > 8:
            a[:, 2:6, 2:min(W, 40)] = (255, 200, 0)
    else:                                                 # steps and stripes
        a = np.zeros((N, H, W, ch), np.uint8)
        p = int(rng.integers(1, 9))
        a[:, :, (np.arange(W) // p) % 2 == 0] = int(rng.integers(1, 256))
        a[:, H // 2:] //= 2
    return np.ascontiguousarray(a)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import tee_optical_flow_amd as T
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    eng = T.DenseFlow()
    bad = 0
    t0 = time.time()
    for c in range(cases):
        small = rng.random() < 0.3
        H = int(rng.integers(1, 40)) if small else int(rng.integers(40, 700))
        W = int(rng.integers(1, 40)) if small else int(rng.integers(40, 900))
        N = int(rng.integers(1, 6))
        ch = int(rng.choice([1, 3, 3]))
        a = draw(rng, N, H, W, ch)
        dt = np.float32 if rng.random() < 0.5 else np.uint8          # the CV_32F map in [0,1] (default hand-over) or the 8-bit map
        got = eng.saliency_frames(a if ch == 3 else a[..., 0], dtype=dt)
        ref = np.stack([O.saliency_fine_grained(f if ch == 3 else f[..., 0], dt) for f in a])
        ok = got.dtype == ref.dtype and np.array_equal(got.view(np.uint32 if dt == np.float32 else np.uint8), ref.view(np.uint32 if dt == np.float32 else np.uint8))
        print(f"case {c}: {N} x {H}x{W}x{ch} {np.dtype(dt).name}: {'ok' if ok else 'FAIL ' + str(int(np.count_nonzero(got != ref))) + ' values differ'}", flush=True)
        bad += not ok
    eng.close()
    print(f"{cases - bad}/{cases} cases identical in {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
