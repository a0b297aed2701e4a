#!/usr/bin/env python3
"""Interleaved A/B of tvl1_iter kernel forms / knobs in ONE process on one MI355X (guide rule 24).
usage: python tools/ab_iter.py [--batch 128] [--rounds 3] variant=0 variant=1,strip_blocks=4096 ..."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--no-profile", action="store_true", help="no HIP events around the launches: wall time only (what the bench's headline measures)")
    ap.add_argument("configs", nargs="*", default=["iter_variant=0", "iter_variant=1"])
    a = ap.parse_args()
    import torch
    import tee_optical_flow_amd as T
    from bench import make_inputs
    B, H, W = a.batch, a.size, a.size
    I0s, I1s = make_inputs(list(range(B)), H, W)
    dev = torch.device("cuda", 0)
    frames = torch.from_numpy(np.concatenate([I0s, I1s])).to(dev)
    flow = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
    ref = None
    eng = T.DenseFlow(max_batch=B)
    p0, p1 = frames.data_ptr(), frames.data_ptr() + B * H * W
    eng.calc_pairs_device(p0, p1, B, H, W, flow.data_ptr())
    res = {c: [] for c in a.configs}
    for r in range(a.rounds):
        for c in a.configs:
            for kv in c.split(","):
                k, v = kv.split("=")
                eng.set_tuning(k, int(v))
            eng.set_profile(0 if a.no_profile else 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = eng.calc_pairs_device(p0, p1, B, H, W, flow.data_ptr())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            f = flow.cpu().numpy()
            if ref is None:
                ref = f
            same = bool(np.array_equal(ref, f))
            res[c].append((dt * 1e3, st["iter_ms"], st["iter_bytes"] / 1e9 / max(st["iter_ms"] / 1e3, 1e-9), st["iter_launches"], same))
    import hashlib
    print("lib", os.environ.get("TEEFLOW_LIB", "default"), "flow sha1", hashlib.sha1(ref.tobytes()).hexdigest()[:16])
    for c, v in res.items():
        v = np.array(v, dtype=np.float64)
        print(f"{c:45s} step ms med {np.median(v[:,0]):8.2f} min {v[:,0].min():8.2f} | iter ms med {np.median(v[:,1]):8.2f} "
              f"| iter GB/s med {np.median(v[:,2]):8.1f} max {v[:,2].max():8.1f} | launches {int(v[0,3])} | identical {bool(v[:,4].all())}")


if __name__ == "__main__":
    main()
