#!/usr/bin/env python3
"""Wide frames (> 1024 px): tile kernel (one iteration per launch) against the full-width strip kernel with 512-thread blocks."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    import tee_optical_flow_amd as T
    from oracle import oracle as O
    from tee_optical_flow_amd.synth import speckle_sequence
    for (H, W, N) in [(720, 1280, 9), (1080, 1920, 9), (768, 1100, 9)]:
        fr = speckle_sequence(H + W, N, H, W)
        ref = None
        for msw in (1024, 2048):
            eng = T.DenseFlow(max_batch=N - 1)
            eng.set_tuning("max_strip_width", msw)
            eng.calc_batch(fr)
            t0 = time.perf_counter(); f = eng.calc_batch(fr); dt = time.perf_counter() - t0
            it = eng.last_iters()
            if ref is None:
                ref = (f.copy(), it.copy())
                o, oit, nl = O.tvl1_calc(fr[0], fr[1], return_iters=True)
                print(f"{H}x{W}: oracle parity of pair 0: {bool(np.array_equal(f[0], o) and np.array_equal(it[0], oit[:nl]))}", flush=True)
            print(f"{H}x{W} max_strip_width {msw}: {(N - 1) / dt:7.1f} pairs/s  identical {bool(np.array_equal(ref[0], f) and np.array_equal(ref[1], it))}", flush=True)
            eng.close()

if __name__ == "__main__":
    main()
