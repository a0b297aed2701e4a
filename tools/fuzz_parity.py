#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity: random sizes, batch sizes and DualTVL1 parameters (all the cv2 setters the engine
supports), flows and executed iteration counts compared bit for bit.  usage: python tools/fuzz_parity.py [cases] [seed] [big]
("big": sizes 161..640 x 201..800 instead of 5..160 x 5..200, fewer pairs per case -- the oracle takes about a second per pair there)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    import tee_optical_flow_amd as T
    from oracle import oracle as O
    from tee_optical_flow_amd.synth import speckle_pairs
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for c in range(cases):
        H = int(rng.integers(5, 161)); W = int(rng.integers(5, 201)); B = int(rng.choice([1, 2, 3, 7, 20, 40, 70]))
        if big:
            H = int(rng.integers(161, 641)); W = int(rng.integers(201, 801)); B = int(rng.choice([1, 2, 9, 20, 33]))
        p = dict(tau=float(rng.choice([0.25, 0.2, 0.1])), lambda_=float(rng.choice([0.15, 0.05, 0.3, 1.0])),
                 theta=float(rng.choice([0.3, 0.2, 0.5])), nscales=int(rng.integers(1, 7)), warps=int(rng.integers(1, 6)),
                 epsilon=float(rng.choice([0.01, 0.02, 0.005, 0.05])), inner_iterations=int(rng.choice([30, 10, 7, 2, 1, 16, 9, 3, 12])),
                 outer_iterations=int(rng.choice([10, 1, 3, 5])), scale_step=float(rng.choice([0.8, 0.55, 0.7, 0.9])),
                 median_filtering=int(rng.choice([5, 3, 1])))
        variant = "cuda" if rng.random() < 0.25 else "cpu"        # TF_VARIANT_CUDA needs an even iteration count
        if variant == "cuda" and (p["inner_iterations"] * p["outer_iterations"]) % 2:
            p["inner_iterations"] += 1
        I0s, I1s = speckle_pairs(range(1000 * c, 1000 * c + B), H, W)
        if rng.random() < 0.3:
            I1s[0] = I0s[0]                                   # identical pair: exact zero flow, stops at once
        eng = T.DenseFlow(max_batch=int(rng.choice([B, max(1, B // 2), 64])), variant=variant, **p)
        if rng.random() < 0.5:
            eng.set_tuning("min_rows_work", 0)                # force the strip kernels even for tiny work
        eng.set_tuning("lanes", int(rng.choice([1, 2])))
        eng.set_tuning("iter_variant", int(rng.choice([2, 2, 1, 0])))      # two iterations per launch on strips / tiles, one per launch, 64x16 tiles
        eng.set_tuning("queue_lanes", int(rng.choice([-1, -1, 0, 2])))       # calls above the capacity: lanes take sub-batches from the queue / contiguous parts
        flows = eng.calc_pairs(I0s, I1s)
        iters = eng.last_iters()
        op = O.default_params(variant=1 if variant == "cuda" else 0)
        for k, v in p.items():
            setattr(op, "lambda_" if k == "lambda_" else k, v)
        ok = True
        for b in sorted(set([0, B - 1, int(rng.integers(0, B))])):
            ref, ref_it, nl = O.tvl1_calc(I0s[b], I1s[b], params=op, return_iters=True)
            same = np.array_equal(flows[b], ref) and np.array_equal(iters[b], ref_it[:nl]) and iters.shape[1] == nl
            if not same:
                ok = False
                print(f"MISMATCH case {c} pair {b}: H={H} W={W} B={B} {p}: {np.sum(flows[b] != ref)} values differ, iters equal "
                      f"{np.array_equal(iters[b], ref_it[:nl])}", flush=True)
        bad += not ok
        eng.close()
        print(f"case {c}: {H}x{W} B={B} scales={p['nscales']} warps={p['warps']} inner={p['inner_iterations']} outer={p['outer_iterations']} "
              f"median={p['median_filtering']} variant={variant} {'ok' if ok else 'FAIL'}", flush=True)
    print(f"{cases - bad}/{cases} cases identical in {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
