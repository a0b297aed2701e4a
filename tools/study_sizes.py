#!/usr/bin/env python3
"""Throughput of one study-sized call (N frames -> N-1 flows through the host-pointer API, PCIe included) at common echo frame sizes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_sequence
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 65
    for algo in ("TVL1", "deepflow"):
        for (H, W) in [(256, 256), (512, 512), (434, 636), (600, 800), (768, 1024), (720, 1280), (1080, 1920)]:
            if algo == "deepflow" and H * W > 800 * 1024:
                continue
            fr = speckle_sequence(H * 3 + W, N, H, W)
            eng = T.DenseFlow(max_batch=N - 1, algo=algo)
            eng.calc_batch(fr)
            ts = []
            for _ in range(2):
                t0 = time.perf_counter(); eng.calc_batch(fr); ts.append(time.perf_counter() - t0)
            print(f"{algo:8s} {H:4d}x{W:<4d} {N} frames: {(N - 1) / min(ts):8.1f} pairs/s  ({min(ts) * 1e3:7.1f} ms per study, {(N - 1) * H * W / min(ts) / 1e6:7.1f} Mpx/s)", flush=True)
            eng.close()

if __name__ == "__main__":
    main()
