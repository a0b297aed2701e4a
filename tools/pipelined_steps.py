#!/usr/bin/env python3
"""What overlapping CONSECUTIVE steps would give (an experiment, not the benchmark's definition of a step): K batches of B pairs, HBM
resident, solved (a) as bench.py does -- one engine, the step split over its lanes, the lanes joined at every step's end -- and (b) by E
engines that each take whole steps in turn from E host threads, so that one step's tail (few pairs still iterating, fine pyramid levels
done) overlaps the next step's start.  Flows of (b) are checked against (a) bit for bit.
usage: python3 tools/pipelined_steps.py [--steps 12] [--batch 128] [--engines 2] [--lanes-each 1]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--engines", type=int, default=2)
    ap.add_argument("--lanes-each", type=int, default=1)
    ap.add_argument("--algo", default="TVL1")
    ap.add_argument("--tuning", default="", help="engine knobs for the in-flight engines, name=value,...")
    a = ap.parse_args()
    from bench import make_inputs
    B, S = a.batch, a.size
    I0s, I1s = make_inputs(list(range(B)), S, S, allow_pool=False)
    import torch
    import tee_optical_flow_amd as T
    dev = torch.device("cuda", 0)
    frames = torch.from_numpy(np.concatenate([I0s, I1s])).to(dev)
    p0, p1 = frames.data_ptr(), frames.data_ptr() + B * S * S

    def run(engines, steps):
        flows = [torch.empty((B, S, S, 2), dtype=torch.float32, device=dev) for _ in engines]
        for e, f in zip(engines, flows):
            e.calc_pairs_device(p0, p1, B, S, S, f.data_ptr())           # warm-up
        torch.cuda.synchronize()
        nxt = [0]
        lock = threading.Lock()

        def worker(e, f):
            while True:
                with lock:
                    k = nxt[0]; nxt[0] += 1
                if k >= steps:
                    return
                e.calc_pairs_device(p0, p1, B, S, S, f.data_ptr())
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(e, f)) for e, f in zip(engines, flows)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dt, flows

    base = T.DenseFlow(max_batch=B, algo=a.algo)
    base.set_tuning("lanes", 2)
    dt0, f0 = run([base], a.steps)
    print(f"{a.algo} one engine, 2 lanes, steps joined  : {dt0 / a.steps * 1e3:7.2f} ms per step, {a.steps * B / dt0:7.1f} pairs/s")
    ref = f0[0].cpu().numpy()
    base.close()
    engs = []
    for _ in range(a.engines):
        e = T.DenseFlow(max_batch=B, algo=a.algo)
        e.set_tuning("lanes", a.lanes_each)
        for kv in filter(None, a.tuning.split(",")):
            k, v = kv.split("=")
            e.set_tuning(k, int(v))
        engs.append(e)
    dt1, f1 = run(engs, a.steps)
    same = all(np.array_equal(f.cpu().numpy(), ref) for f in f1)
    print(f"{a.algo} {a.engines} engines x {a.lanes_each} lane(s), whole steps in turn: {dt1 / a.steps * 1e3:7.2f} ms per step, {a.steps * B / dt1:7.1f} pairs/s, flows identical: {same}")
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
