#!/usr/bin/env python3
"""A/B of DeepFlow engine knobs on one study-sized call: python tools/ab_study.py H W "knob=v,knob=v" ["..."]  (65 frames, host-pointer API)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_sequence
    H, W = int(sys.argv[1]), int(sys.argv[2])
    N = 65
    fr = speckle_sequence(H * 3 + W, N, H, W)
    ref = None
    for cfg in sys.argv[3:]:
        eng = T.DenseFlow(max_batch=N - 1, algo="deepflow")
        for kv in filter(None, cfg.split(",")):
            k, v = kv.split("=")
            eng.set_tuning(k, int(v))
        out = np.asarray(eng.calc_batch(fr))
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); eng.calc_batch(fr); ts.append(time.perf_counter() - t0)
        same = ref is None or np.array_equal(ref, out)
        ref = out if ref is None else ref
        print(f"{H}x{W} {cfg:40s} {(N - 1) / min(ts):8.1f} pairs/s  coop launches {eng.counter('coop_launches')}  identical {same}", flush=True)
        eng.close()

if __name__ == "__main__":
    main()
