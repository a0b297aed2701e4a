#!/usr/bin/env python3
"""Host-pointer entry point (PCIe in and out included): pairs/s for a few sub-batch counts and for a pageable destination.
usage: python tools/pcie_path.py [--batch 128]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--size", type=int, default=512)
    a = ap.parse_args()
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd import _lib
    from bench import make_inputs
    B, H, W = a.batch, a.size, a.size
    I0s, I1s = make_inputs(list(range(B)), H, W)
    eng = T.DenseFlow(max_batch=B)
    ref = None
    for sb in (1, 2, 3, 4, 8):
        eng.set_tuning("sub_batches", sb)
        eng.calc_pairs(I0s, I1s)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            f = eng.calc_pairs(I0s, I1s)
            ts.append(time.perf_counter() - t0)
        st = eng.last_stats
        if ref is None:
            ref = f.copy()
        print(f"pinned result, sub_batches={sb}: {B / min(ts):8.1f} pairs/s  (device {st['ms_device']:.1f} ms, d2h {st['ms_d2h']:.1f} ms, "
              f"h2d {st['ms_h2d']:.1f} ms; identical {np.array_equal(ref, f)})", flush=True)
        del f
    out = np.empty((B, H, W, 2), np.float32)
    out[...] = 0
    st = _lib.TfStats()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        _lib.check(eng._L.tf_calc_pairs(eng._h, I0s.ctypes.data, I1s.ctypes.data, B, H, W, out.ctypes.data, C.byref(st)), eng._h)
        ts.append(time.perf_counter() - t0)
    print(f"pageable destination (touched):   {B / min(ts):8.1f} pairs/s  identical {np.array_equal(ref, out)}", flush=True)
    t0 = time.perf_counter()
    fresh = np.empty((B, H, W, 2), np.float32)
    _lib.check(eng._L.tf_calc_pairs(eng._h, I0s.ctypes.data, I1s.ctypes.data, B, H, W, fresh.ctypes.data, C.byref(st)), eng._h)
    print(f"pageable destination (fresh np.empty, first touch): {B / (time.perf_counter() - t0):8.1f} pairs/s", flush=True)


if __name__ == "__main__":
    main()
