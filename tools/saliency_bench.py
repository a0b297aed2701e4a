#!/usr/bin/env python3
"""The no_saliency=False preprocessing (reference calculate_optical_flow.py:559-560, :586) by itself: tf_saliency_frames_f32 on a study's
worth of 512x512 RGB frames.  Prints wall ms per frame (host frames in, float maps out: PCIe both ways) and the device time of the eight
kernels per frame (HIP events inside the library) against the bytes they have to move.
usage: python3 tools/saliency_bench.py [--frames 65] [--size 512] [--reps 10]      (under rocprofv3 --kernel-trace --stats for per-kernel shares)"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# compulsory HBM bytes per pixel of the eight passes (each plane read / written once): gray 3+1, blur 2 x (1+1), row prefix 1+4, integral 4+4,
# scales 1+4 (+ the 24 gathered corners, which a cache serves) +2+2, mix_scales 4+2, mix_onoff 2+4
BYTES_PER_PX = 3 + 1 + 2 * 2 + 5 + 8 + 9 + 6 + 6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=65)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    from tee_optical_flow_amd.synth import speckle_sequence
    import tee_optical_flow_amd as T
    seq = speckle_sequence(3, a.frames, a.size, a.size)
    rgb = np.ascontiguousarray(np.repeat(seq[..., None], 3, axis=3))
    eng = T.DenseFlow()
    eng.saliency_frames(rgb)
    wall, dev = [], []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        eng.saliency_frames(rgb)
        wall.append(time.perf_counter() - t0)
        dev.append(eng.counter("saliency_kernel_us") * 1e-6)
    w, d = float(np.median(wall)), float(np.median(dev))
    px = a.frames * a.size * a.size
    print(f"saliency maps of {a.frames} frames {a.size}x{a.size} RGB: {w / a.frames * 1e3:.3f} ms per frame end to end (PCIe both ways), "
          f"{d / a.frames * 1e6:.1f} us per frame on the device = {px * BYTES_PER_PX / d / 1e9:.0f} GB/s of the {BYTES_PER_PX} B per pixel the passes must move "
          f"({px * BYTES_PER_PX / d / 8e12:.3f} of the 8 TB/s peak)")
    eng.close()


if __name__ == "__main__":
    main()
