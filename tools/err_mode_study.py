#!/usr/bin/env python3
"""How much does deviation D1 of oracle/tvl1_oracle.c (exact integer convergence sum instead of upstream's float raster-order
sum) hide?  Runs the CPU oracle both ways -- err_mode 0 (exact, what the GPU engine matches bit for bit) and err_mode 1 (one
float accumulator in raster order, upstream's form) -- on the 128 benchmark pairs (512x512, speckle-warp v1 seeds 0..127) and on
randomised small cases with random parameters, and reports how many (level, warp) stages stop at a different iteration and
what that does to the flow (EPE between the two results).  This is the floor any comparison with real
cv2.optflow.createOptFlow_DualTVL1() (reference calculate_optical_flow.py:577-578, 642) will see from this one deviation.
CPU only.   usage: python tools/err_mode_study.py [--pairs 128] [--fuzz 500] [--out profiles/r03_err_mode_study.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def compare(O, I0, I1, **over):
    p0 = O.default_params(err_mode=0, **over)
    p1 = O.default_params(err_mode=1, **over)
    f0, it0, nl = O.tvl1_calc(I0, I1, p0, return_iters=True)
    f1, it1, nl1 = O.tvl1_calc(I0, I1, p1, return_iters=True)
    assert nl == nl1
    a, b = it0[:nl, :, 0], it1[:nl, :, 0]
    epe = np.sqrt(((f0 - f1) ** 2).sum(-1))
    return {"stages": int(a.size), "stages_differ": int((a != b).sum()), "max_iter_delta": int(np.abs(a - b).max()),
            "identical": bool(np.array_equal(f0, f1)), "epe_mean": float(epe.mean()), "epe_max": float(epe.max())}


def summarise(rows):
    n = len(rows)
    if not n:
        return {}
    d = [r for r in rows if not r["identical"]]
    return {"cases": n, "stages": sum(r["stages"] for r in rows), "stages_with_a_different_stop": sum(r["stages_differ"] for r in rows),
            "cases_with_a_different_stop": sum(1 for r in rows if r["stages_differ"]), "cases_with_a_different_flow": len(d),
            "largest_iteration_delta": max(r["max_iter_delta"] for r in rows),
            "mean_epe_over_all_cases": float(np.mean([r["epe_mean"] for r in rows])),
            "mean_epe_over_differing_cases": float(np.mean([r["epe_mean"] for r in d])) if d else 0.0,
            "worst_case_mean_epe": max(r["epe_mean"] for r in rows), "worst_case_max_epe": max(r["epe_max"] for r in rows)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--fuzz", type=int, default=500)
    ap.add_argument("--out", default="profiles/r03_err_mode_study.json")
    a = ap.parse_args()
    from oracle import oracle as O
    from tee_optical_flow_amd.synth import speckle_pair
    O.build()
    O.set_num_threads(O.effective_cpus())
    t0 = time.time()
    bench_rows = []
    for s in range(a.pairs):
        I0, I1, _ = speckle_pair(s, 512, 512)
        bench_rows.append(compare(O, I0, I1))
        if s % 16 == 15:
            print(f"benchmark pairs {s + 1}/{a.pairs}  {time.time() - t0:.0f} s", flush=True)
    rng = np.random.default_rng(2026)
    fuzz_rows = []
    for c in range(a.fuzz):
        H = int(rng.integers(24, 161)); W = int(rng.integers(24, 201))
        over = dict(tau=float(rng.choice([0.25, 0.2, 0.1])), lambda_=float(rng.choice([0.15, 0.05, 0.3, 1.0])), theta=float(rng.choice([0.3, 0.2, 0.5])),
                    nscales=int(rng.integers(1, 6)), warps=int(rng.integers(1, 6)), epsilon=float(rng.choice([0.01, 0.02, 0.005, 0.05])),
                    inner_iterations=int(rng.choice([30, 10, 7, 16])), outer_iterations=int(rng.choice([10, 3, 5])),
                    median_filtering=int(rng.choice([5, 3, 1])))
        I0, I1, _ = speckle_pair(10000 + c, H, W)
        fuzz_rows.append(compare(O, I0, I1, **over))
    out = {"what": "oracle err_mode 0 (exact integer convergence sum, deviation D1; what the HIP engine reproduces) vs err_mode 1 (upstream's single "
                   "float accumulator in raster order); CPU oracle only",
           "benchmark_pairs_512": summarise(bench_rows), "fuzz_small_random_parameters": summarise(fuzz_rows),
           "seconds": time.time() - t0, "threads": O.effective_cpus()}
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
