#!/usr/bin/env python3
"""Where a block of k_df_sor_rt_coop spends its time: s_memrealtime stamps of wave 0 of three blocks of the LAST launch of a solve (level 0).
Needs the library built with -DTF_COOP_TIMING:  TEEFLOW_LIB=tools/microbench/libteeflow_timing.so python3 tools/coop_timing.py [pairs] [S] [lanes]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd import _lib
    from tee_optical_flow_amd.synth import speckle_pairs
    I0s, I1s = speckle_pairs(range(B), 512, 512)
    e = T.DenseFlow(algo="deepflow", max_batch=B)
    e.set_tuning("lanes", lanes)
    e.set_tuning("sor_coop_s", S)
    e.calc_pairs(I0s, I1s)
    e.calc_pairs(I0s, I1s)
    L = _lib.load()
    out = np.zeros((4, 64), np.uint64)
    L.tf_dbg_coop_times.argtypes = [C.c_void_p]
    assert L.tf_dbg_coop_times(out.ctypes.data_as(C.c_void_p)) == 0
    names = ["sweeps", "stores acked (wave 0)", "all waves' stores acked", "neighbours met", "(barrier)+reload landed"]
    for slot, what in enumerate(["first block", "middle block of pair 0", "last block"]):
        t = out[slot].astype(np.int64)
        if t[0] == 0:
            continue
        print(f"{what}: load + set-up {(t[1] - t[0]) / 100:.2f} us")
        i = 2
        ph = 1
        while i + 1 < 64 and t[i + 1] > 0:
            seg = [(t[i + 1] - t[i]) / 100]
            k = i + 1
            while k + 1 < min(i + 6, 64) and t[k + 1] > 0:
                seg.append((t[k + 1] - t[k]) / 100)
                k += 1
            gap = (t[i] - t[i - 1]) / 100
            print(f"  phase {ph}: (+{gap:.2f} before) " + ", ".join(f"{n} {v:.2f}" for n, v in zip(names, seg)))
            i += 6
            ph += 1
        last = max(t)
        print(f"  total {(last - t[0]) / 100:.2f} us")
    e.close()


if __name__ == "__main__":
    main()
