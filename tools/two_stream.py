#!/usr/bin/env python3
"""Does splitting a 128-pair step over K engines (own HIP streams, own host threads) hide the lock-step tails?"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import tee_optical_flow_amd as T
    from bench import make_inputs
    B, H, W = 128, 512, 512
    I0s, I1s = make_inputs(list(range(B)), H, W)
    dev = torch.device("cuda", 0)
    for K in (1, 2, 4):
        n = B // K
        engs, frs, fls = [], [], []
        for k in range(K):
            fr = torch.from_numpy(np.concatenate([I0s[k * n:(k + 1) * n], I1s[k * n:(k + 1) * n]])).to(dev)
            frs.append(fr)
            fls.append(torch.empty((n, H, W, 2), dtype=torch.float32, device=dev))
            engs.append(T.DenseFlow(max_batch=n))

        def run(k):
            p0 = frs[k].data_ptr()
            engs[k].calc_pairs_device(p0, p0 + n * H * W, n, H, W, fls[k].data_ptr())

        def step():
            ts = [threading.Thread(target=run, args=(k,)) for k in range(K)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        print(f"K={K} engines x {n} pairs: {dt * 1e3:.2f} ms/step  {B / dt:.1f} pairs/s", flush=True)
        for e in engs:
            e.close()


if __name__ == "__main__":
    main()
