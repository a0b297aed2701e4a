#!/usr/bin/env python3
"""One DeepFlow solve of B pairs on one lane, no event profiling: the program rocprofv3 --kernel-trace wraps for tools/gap_trace.py.
usage: python3 tools/df_trace_run.py [B] [size] [tuning k=v,k=v] [lanes]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    tuning = sys.argv[3] if len(sys.argv) > 3 else ""
    lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    from bench import make_inputs
    I0s, I1s = make_inputs(list(range(B)), size, size, allow_pool=False)
    import torch
    import tee_optical_flow_amd as T
    dev = torch.device("cuda", 0)
    frames = torch.from_numpy(np.concatenate([I0s, I1s])).to(dev)
    flow = torch.empty((B, size, size, 2), dtype=torch.float32, device=dev)
    eng = T.DenseFlow(max_batch=B, algo="deepflow")
    eng.set_tuning("lanes", lanes)
    for kv in filter(None, tuning.split(",")):
        k, v = kv.split("=")
        eng.set_tuning(k, int(v))
    p0, p1 = frames.data_ptr(), frames.data_ptr() + B * size * size
    eng.calc_pairs_device(p0, p1, B, size, size, flow.data_ptr())      # warm-up (allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.calc_pairs_device(p0, p1, B, size, size, flow.data_ptr())
    torch.cuda.synchronize()
    print(f"deepflow B={B} {size}x{size} lanes={lanes} tuning='{tuning}': {1e3 * (time.perf_counter() - t0):.2f} ms wall")


if __name__ == "__main__":
    main()
