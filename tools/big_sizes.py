#!/usr/bin/env python3
"""Parity at realistic echo frame sizes (strip kernels up to 1024 px wide, the tile kernel beyond), sector-masked frames,
sequence mode: GPU flows and executed iteration counts against the oracle, bit for bit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sector(H, W):
    yy, xx = np.mgrid[0:H, 0:W]
    ang = np.arctan2(xx - W / 2, yy + H * 0.05)
    return (np.abs(ang) < 0.7) & (np.hypot(xx - W / 2, yy + H * 0.05) < H * 0.98)


def main():
    import tee_optical_flow_amd as T
    from oracle import oracle as O
    from tee_optical_flow_amd.synth import speckle_sequence
    bad = 0
    for (H, W, N) in [(600, 800, 4), (434, 636, 5), (708, 1016, 3), (768, 1024, 3), (720, 1280, 3), (1080, 1920, 2), (333, 1025, 3)]:
        t0 = time.time()
        fr = speckle_sequence(H + W, N, H, W)
        fr = np.where(sector(H, W)[None], fr, 0).astype(np.uint8)
        eng = T.DenseFlow(max_batch=8)
        flows = eng.calc_batch(fr)
        iters = eng.last_iters()
        ok = True
        for i in sorted(set([0, N - 2])):
            ref, ref_it, nl = O.tvl1_calc(fr[i], fr[i + 1], return_iters=True)
            ok &= bool(np.array_equal(flows[i], ref) and np.array_equal(iters[i], ref_it[:nl]))
        bad += not ok
        eng.close()
        print(f"{H}x{W} sequence of {N}: {'ok' if ok else 'FAIL'}  ({time.time() - t0:.1f} s)", flush=True)
    # DeepFlow on the same kind of frames: outside the sector everything is exactly zero, so the SOR works next to flat regions where
    # du, dv decay geometrically into the denormal range -- the regime the pre-scaled division of k_df_sor_rt has to get right
    for (H, W, N) in [(434, 636, 3), (600, 800, 3), (333, 1025, 2), (512, 512, 3)]:
        t0 = time.time()
        fr = speckle_sequence(2 * H + W, N, H, W)
        fr = np.where(sector(H, W)[None], fr, 0).astype(np.uint8)
        eng = T.DenseFlow(max_batch=8, algo="deepflow")
        flows = eng.calc_batch(fr)
        ok = all(bool(np.array_equal(flows[i], O.deepflow_calc(fr[i], fr[i + 1]))) for i in sorted(set([0, N - 2])))
        bad += not ok
        eng.close()
        print(f"DeepFlow {H}x{W} sequence of {N}: {'ok' if ok else 'FAIL'}  ({time.time() - t0:.1f} s)", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
