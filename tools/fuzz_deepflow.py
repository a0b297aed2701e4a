#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity for the DeepFlow path: random sizes and batch sizes, flows compared bit for bit.
usage: python tools/fuzz_deepflow.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import tee_optical_flow_amd as T
    from oracle import oracle as O
    from tee_optical_flow_amd.synth import speckle_pairs
    rng = np.random.default_rng(seed)
    bad = 0
    ncoop = 0
    t0 = time.time()
    for c in range(cases):
        # half of the cases are large enough for co-resident launches with several launches per level
        big = rng.random() < 0.65
        H = int(rng.integers(200, 641)) if big else int(rng.integers(26, 220))
        W = int(rng.integers(260, 641)) if big else int(rng.integers(26, 300))
        B = int(rng.choice([2, 5, 9]) if big else rng.choice([1, 2, 5, 33]))
        I0s, I1s = speckle_pairs(range(500 * c, 500 * c + B), H, W)
        if rng.random() < 0.25:
            I1s[0] = I0s[0]
        eng = T.DenseFlow(algo="deepflow", max_batch=int(rng.choice([B, max(1, B // 2)])))
        shape = 3 if big else int(rng.choice([1, 2, 3, 3, 3]))   # register-tile SOR: 16 x 4 / 8 x 4 bands / chosen per launch (the co-resident form needs 3)
        fuse = int(rng.choice([0, 1, 2, 3, 4, 5, 5, 6, 7, 8]))   # sweeps per launch (0: one colour per launch)
        eng.set_tuning("sor_rt_shape", shape)
        eng.set_tuning("sor_fuse", fuse)
        eng.set_tuning("df_fuse_ds", int(rng.choice([0, 2, 2])))
        coop = int(rng.choice([1, 1, 2, 2, 3]) if big else rng.choice([0, 1, 1, 2, 2, 3]))            # co-resident regions (2: 128 x 64 whatever the batch size, 3: 128 x 32 for small batches), exchanging every coop_s sweeps
        coop_s = int(rng.choice([1, 2, 3, 4, 5, 5, 6, 7]))
        eng.set_tuning("sor_coop", coop)
        eng.set_tuning("sor_coop_s", coop_s)
        flows = eng.calc_pairs(I0s, I1s)
        ok = True
        for b in sorted(set([0, B - 1])):
            ref = O.deepflow_calc(I0s[b], I1s[b])
            if not np.array_equal(flows[b], ref):
                ok = False
                print(f"MISMATCH case {c} pair {b}: {H}x{W} B={B}: {np.sum(flows[b] != ref)} values differ, max {np.abs(flows[b] - ref).max()}", flush=True)
        ok = ok and eng.counter("coop_aborts") == 0
        print(f"case {c}: {H}x{W} B={B} shape={shape} fuse={fuse} coop={coop}/{coop_s} ({eng.counter('coop_launches')} launches) {'ok' if ok else 'FAIL'}", flush=True)
        ncoop += eng.counter('coop_launches') > 0
        eng.close()
        bad += not ok
    print(f"{cases - bad}/{cases} cases identical in {time.time() - t0:.0f} s; {ncoop} of them reached co-resident launches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
