#!/usr/bin/env python3
"""The library's lane queue against the forms it replaces, same box, same pairs, interleaved (round 5).
Every form solves TOTAL = steps x 128 pairs @512^2 resident in HBM; flows are compared bit for bit with form (a).
  a  one engine, 128-pair calls, each split over two lanes and joined            (rounds 1-3: "steps_joined")
  b  three engines driven by three Python threads, whole 128-pair calls in turn  (round 4's bench.py / EnginePool)
  c  ONE engine, tf_submit_pairs_device of 128-pair jobs, E in flight            (the library's lanes take whole sub-batches)
  d  ONE engine, ONE synchronous tf_calc_pairs_device per K x 128 pairs          (the same queue behind one call of the boundary)
usage: python3 tools/queue_forms.py [--steps 24] [--calls 384,768,1024] [--rounds 2] [--algo TVL1] [--tuning k=v,...]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--calls", default="384,768,1024", help="pairs per synchronous call of form d")
    ap.add_argument("--in-flight", default="2,3,4", help="jobs in flight of form c")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--algo", default="TVL1")
    ap.add_argument("--tuning", default="")
    ap.add_argument("--skip-b", action="store_true")
    ap.add_argument("--no-pool", action="store_true")
    ap.add_argument("--distinct", action="store_true", help="384 DISTINCT pairs (seeds 0..383, what bench.py solves per step) instead of the same 128 three times: "
                                                            "the sub-batches of a call then differ in length")
    a = ap.parse_args()
    from bench import make_inputs
    B, S = 128, a.size
    calls = [int(c) for c in a.calls.split(",") if c]
    PM = max(calls + [B])
    I0s, I1s = make_inputs(list(range(B)), S, S, allow_pool=False)
    if a.distinct:
        D0, D1 = make_inputs(list(range(3 * B)), S, S, allow_pool=not a.no_pool)
    import torch
    import tee_optical_flow_amd as T
    dev = torch.device("cuda", 0)
    reps = (PM + B - 1) // B
    # [I0 x reps | I1 x reps]: every 128-pair slice is the same 128 pairs, so every form does the same work per pair
    if a.distinct:                                                  # pairs k, k + 384, ... are the same pair
        big0, big1 = np.tile(D0, (-(-reps // 3), 1, 1))[:reps * B], np.tile(D1, (-(-reps // 3), 1, 1))[:reps * B]
    else:
        big0, big1 = np.tile(I0s, (reps, 1, 1)), np.tile(I1s, (reps, 1, 1))
    frames = torch.from_numpy(np.concatenate([big0, big1])).to(dev)
    p0, p1 = frames.data_ptr(), frames.data_ptr() + reps * B * S * S
    big = torch.empty((reps * B, S, S, 2), dtype=torch.float32, device=dev)
    ring = [torch.empty((B, S, S, 2), dtype=torch.float32, device=dev) for _ in range(8)]
    tuning = [kv.split("=") for kv in a.tuning.split(",") if kv]

    def engine(**kw):
        e = T.DenseFlow(max_batch=B, algo=a.algo, **kw)
        for k, v in tuning:
            e.set_tuning(k, int(v))
        return e

    one = engine()
    three = [] if a.skip_b else [engine() for _ in range(3)]
    for e in three:
        e.set_tuning("lanes", 1)
    ref = None

    NSUB = 3 if a.distinct else 1
    off = lambda k: (k % NSUB) * B * S * S

    def form_a(steps):
        for k in range(steps):
            one.calc_pairs_device(p0 + off(k), p1 + off(k), B, S, S, ring[k % NSUB].data_ptr())
        return ring[0]

    def form_b(steps):
        nxt, lock = [0], threading.Lock()

        def worker(i):
            while True:
                with lock:
                    k = nxt[0]; nxt[0] += 1
                if k >= steps:
                    return
                three[i].calc_pairs_device(p0 + off(k), p1 + off(k), B, S, S, ring[3 + i].data_ptr())
                if k % NSUB == 0:
                    ring[0].copy_(ring[3 + i])
        th = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return ring[0]

    def form_c(steps, E):
        tk = {}
        for k in range(steps + E):
            if k - E >= 0:
                one.wait(tk.pop(k - E))
            if k < steps:
                tk[k] = one.submit_pairs_device(p0 + off(k), p1 + off(k), B, S, S, ring[k % (E + 1)].data_ptr())
        last0 = max(k for k in range(steps) if k % NSUB == 0)
        return ring[last0 % (E + 1)]

    def form_d(steps, P):
        n = max(1, steps * B // P)
        for _ in range(n):
            one.calc_pairs_device(p0, p1, P, S, S, big.data_ptr())
        return big[:B], n * P

    def timed(fn, *args):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*args)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r

    # warm-up: every engine and every lane allocates
    form_a(1)
    ref = ring[0].cpu().numpy()
    if three:
        form_b(3)
    form_c(6, 3)
    for P in calls:
        form_d(1, P)
    print(f"{a.algo} {S}x{S}, {a.steps} x {B} pairs per form, queue lanes {one.counter('queue_lanes')}, tuning '{a.tuning}'")
    for r in range(a.rounds):
        dt, f = timed(form_a, a.steps)
        print(f"  a  one engine, 2 lanes, joined per 128-pair call     : {a.steps * B / dt:7.1f} pairs/s  identical {np.array_equal(f.cpu().numpy(), ref)}")
        if three:
            dt, f = timed(form_b, a.steps)
            print(f"  b  3 engines, 3 Python threads                       : {a.steps * B / dt:7.1f} pairs/s  identical {np.array_equal(f.cpu().numpy(), ref)}")
        for E in [int(x) for x in a.in_flight.split(",") if x]:
            dt, f = timed(form_c, a.steps, E)
            print(f"  c  one engine, tf_submit 128-pair jobs, {E} in flight   : {a.steps * B / dt:7.1f} pairs/s  identical {np.array_equal(f.cpu().numpy(), ref)}")
        for P in calls:
            dt, (f, n) = timed(form_d, a.steps, P)
            same = np.array_equal(f.cpu().numpy(), ref) and (a.distinct or np.array_equal(big[P - B:P].cpu().numpy(), ref))
            print(f"  d  one engine, ONE synchronous call per {P:5d} pairs   : {n / dt:7.1f} pairs/s  identical {same}")
    print(f"  lane / twin streams: tried and dropped {one.counter('stream_retries')}, solve streams serialised on a shared hardware queue: {bool(one.counter('streams_serialised') & 1)}, "
          f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'unset (4)')}")
    one.close()
    for e in three:
        e.close()


if __name__ == "__main__":
    main()
