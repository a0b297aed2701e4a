#!/usr/bin/env python3
"""Diagnostic (round 5): the joined 128-pair form against the queue on the bench's 384 distinct pairs, sub-batch by sub-batch, with the
executed iteration counts of each sub-batch -- is the joined form's low rate on distinct data the data (a slow pair holds a 64-pair lane)?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from bench import make_inputs
    B, S = 128, 512
    I0s, I1s = make_inputs(list(range(3 * B)), S, S, allow_pool=True)
    import torch
    import tee_optical_flow_amd as T
    dev = torch.device("cuda", 0)
    fr = torch.from_numpy(np.concatenate([I0s, I1s])).to(dev)
    npx = S * S
    p0, p1 = fr.data_ptr(), fr.data_ptr() + 3 * B * npx
    out = torch.empty((3 * B, S, S, 2), dtype=torch.float32, device=dev)
    eng = T.DenseFlow(max_batch=B)
    def sub(c, n=8):
        a, b, o = p0 + c * B * npx, p1 + c * B * npx, out.data_ptr() + c * B * npx * 8
        eng.calc_pairs_device(a, b, B, S, S, o)
        it = eng.last_iters()[..., 0].sum(axis=(1, 2))
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            eng.calc_pairs_device(a, b, B, S, S, o)
        torch.cuda.synchronize(); return n * B / (time.perf_counter() - t), it
    for lanes in (2, 1):
        eng.set_tuning("lanes", lanes)
        for c in range(3):
            r, it = sub(c)
            st = eng.last_iters()[..., 0]                      # [pairs, levels, warps]
            print(f"lanes {lanes} sub-batch {c} (seeds {c * B}..{c * B + B - 1}): {r:7.1f} pairs/s; inner iterations per pair mean {it.mean():.0f} max {it.max()} "
                  f"; longest stage of the batch per (level, warp): mean {st.max(axis=0).mean():.1f} vs mean pair {st.mean():.1f}")
    eng.set_tuning("lanes", 2)
    eng.calc_pairs_device(p0, p1, 3 * B, S, S, out.data_ptr())
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(4):
        eng.calc_pairs_device(p0, p1, 3 * B, S, S, out.data_ptr())
    torch.cuda.synchronize(); print("queue, one call of 384:", round(4 * 3 * B / (time.perf_counter() - t), 1))
    eng.close()


if __name__ == "__main__":      # (make_inputs starts a spawn pool: the guard is not optional)
    main()
