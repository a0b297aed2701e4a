#!/usr/bin/env python3
"""DeepFlow throughput for several sor_fuse settings in one process (run as a FILE: input generation uses a spawn pool)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import tee_optical_flow_amd as T
    from bench import make_inputs
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    fuses = [int(x) for x in sys.argv[2:]] or [0, 2, 5]
    I0s, I1s = make_inputs(list(range(B)), 512, 512)
    fr = torch.from_numpy(np.concatenate([I0s, I1s])).cuda()
    fl = torch.empty((B, 512, 512, 2), dtype=torch.float32, device="cuda")
    e = T.DenseFlow(algo="deepflow", max_batch=B)
    for kv in filter(None, os.environ.get("DF_TUNING", "").split(",")):     # e.g. DF_TUNING=sor_rt_shape=0
        k, v = kv.split("=")
        e.set_tuning(k, int(v))
    p0 = fr.data_ptr()
    p1 = p0 + B * 512 * 512
    ref = None
    for f in fuses:
        e.set_tuning("sor_fuse", f)
        e.calc_pairs_device(p0, p1, B, 512, 512, fl.data_ptr())
        t = time.perf_counter()
        e.calc_pairs_device(p0, p1, B, 512, 512, fl.data_ptr())
        dt = time.perf_counter() - t
        out = fl.cpu().numpy()
        if ref is None:
            ref = out
        print(f"sor_fuse {f}: {dt / B * 1e3:.3f} ms/pair  {B / dt:.1f} pairs/s  identical {bool(np.array_equal(ref, out))}", flush=True)


if __name__ == "__main__":
    main()
