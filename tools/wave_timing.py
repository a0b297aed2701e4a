#!/usr/bin/env python3
"""When do the waves of one full k_iter2_wave launch (level 0, every pair active) start, finish their prologue and end?
Needs the library built with -DTF_WAVE_TIMING:
  hipcc <flags of csrc/Makefile> -DTF_WAVE_TIMING csrc/teeflow.hip -o tools/microbench/libteeflow_wtiming.so
  TEEFLOW_LIB=tools/microbench/libteeflow_wtiming.so python3 tools/wave_timing.py [pairs] [tuning]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    tuning = sys.argv[2] if len(sys.argv) > 2 else "iter_variant=5,wave_minrows=4"
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd import _lib
    from bench import make_inputs
    I0s, I1s = make_inputs(list(range(B)), 512, 512, allow_pool=False)
    e = T.DenseFlow(max_batch=B)
    e.set_tuning("lanes", 1)
    for kv in filter(None, tuning.split(",")):
        k, v = kv.split("=")
        e.set_tuning(k, int(v))
    e.calc_pairs(I0s, I1s)
    e.calc_pairs(I0s, I1s)
    L = _lib.load()
    out = np.zeros((4096, 4), np.uint64)
    L.tf_dbg_wave_times.argtypes = [C.c_void_p]
    assert L.tf_dbg_wave_times(out.ctypes.data_as(C.c_void_p)) == 0
    t = out.astype(np.int64)
    on = t[:, 2] > 0
    n = int(on.sum())
    t0 = t[on, 0].min()
    st, pr, en, rows = (t[on, 0] - t0) / 100.0, (t[on, 1] - t0) / 100.0, (t[on, 2] - t0) / 100.0, t[on, 3]
    print(f"{n} waves with work, rows per strip {np.unique(rows)}; launch spans {en.max():.1f} us from the first wave's entry")
    q = lambda a: " ".join(f"{np.percentile(a, p):7.1f}" for p in (0, 5, 25, 50, 75, 95, 100))
    print("percentiles (us)      min      5      25      50      75      95     max")
    print("  entry            ", q(st))
    print("  prologue done    ", q(pr))
    print("  prologue length  ", q(pr - st))
    print("  end              ", q(en))
    print("  march length     ", q(en - pr))
    busy = (en - st).sum() / (n * en.max())
    print(f"mean wave-slot occupancy over the launch: {busy:.3f}   (sum of wave lifetimes / (waves x span))")
    idle_tail = (en.max() - en).mean()
    print(f"mean idle tail per wave {idle_tail:.1f} us = {idle_tail / en.max():.3f} of the launch")
    if hasattr(L, "tf_dbg_step_times"):
        st2 = np.zeros((2, 96), np.uint64)
        L.tf_dbg_step_times.argtypes = [C.c_void_p]
        if L.tf_dbg_step_times(st2.ctypes.data_as(C.c_void_p)) == 0:
            for w in range(2):
                tt = st2[w].astype(np.int64)
                tt = tt[tt > 0]
                if len(tt) > 2:
                    d = np.diff(tt) / 100.0
                    print(f"wave {w}: {len(d)} row steps (us): " + " ".join(f"{x:.1f}" for x in d))
    e.close()


if __name__ == "__main__":
    main()
