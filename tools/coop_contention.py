#!/usr/bin/env python3
"""Two PROCESSES on one GPU, each solving DeepFlow batches with the co-resident SOR form and each counting on every CU: the one case in
which a launch of k_df_sor_rt_coop can really fail to get its regions resident.  Every wait in the kernel is bounded, so the worst that
may happen is an abort + the call repeated with the tiled form; results must be bit-exact either way.
usage: python tools/coop_contention.py [pairs] [size] [rounds]        (parent; starts two children of itself)"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(tag, B, N, rounds, start_at):
    import tee_optical_flow_amd as T
    from oracle import oracle as O
    from tee_optical_flow_amd.synth import speckle_pairs
    I0s, I1s = speckle_pairs(range(7 * tag, 7 * tag + B), N, N)
    ref0 = O.deepflow_calc(I0s[0], I1s[0])
    refl = O.deepflow_calc(I0s[B - 1], I1s[B - 1])
    e = T.DenseFlow(algo="deepflow", max_batch=B)
    e.set_tuning("sor_coop", 2)
    e.calc_pairs(I0s[:1], I1s[:1])
    bad = 0
    while time.time() < start_at:            # both children start solving at the same moment
        time.sleep(0.001)
    t0 = time.time()
    for r in range(rounds):
        fl = e.calc_pairs(I0s, I1s)
        ok = np.array_equal(fl[0], ref0) and np.array_equal(fl[B - 1], refl)
        bad += not ok
        if r % 10 == 0 or not ok or r == rounds - 1:
            print(f"child {tag} round {r}: {'ok' if ok else 'MISMATCH'}  coop launches {e.counter('coop_launches')} aborts {e.counter('coop_aborts')} "
                  f"disabled {e.counter('coop_disabled')}  t={time.time() - t0:.2f}s", flush=True)
        if e.counter("coop_disabled"):
            e.set_tuning("sor_coop", 2)          # re-arm: keep provoking
    e.close()
    sys.exit(1 if bad else 0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]))
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    start_at = time.time() + 45.0
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(k), str(B), str(N), str(rounds), repr(start_at)]) for k in (1, 2)]
    rc = [p.wait(timeout=600) for p in ps]
    print("children exit codes", rc)
    sys.exit(max(rc))


if __name__ == "__main__":
    main()
