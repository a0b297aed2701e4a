#!/usr/bin/env python3
"""Where does tvl1_iter time go in a lock-step batch?  One profiled single-lane solve of B pairs; every launch is tagged
with (level, warp, first iteration); the number of pairs still iterating at that launch follows from the executed
iteration counts.  usage: python tools/launch_profile.py [--batch 64] [--size 512]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--tuning", default="")
    ap.add_argument("--k", type=int, default=2, help="iterations per launch of the form under test")
    ap.add_argument("--lanes", type=int, default=1)
    a = ap.parse_args()
    import torch
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd import _lib
    from bench import make_inputs
    B, H, W = a.batch, a.size, a.size
    I0s, I1s = make_inputs(list(range(B)), H, W)
    dev = torch.device("cuda", 0)
    frames = torch.from_numpy(np.concatenate([I0s, I1s])).to(dev)
    flow = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
    eng = T.DenseFlow(max_batch=B)
    eng.set_tuning("lanes", a.lanes)
    for kv in filter(None, a.tuning.split(",")):
        k, v = kv.split("=")
        eng.set_tuning(k, int(v))
    p0, p1 = frames.data_ptr(), frames.data_ptr() + B * H * W
    eng.calc_pairs_device(p0, p1, B, H, W, flow.data_ptr())
    eng.set_profile(1)
    st = eng.calc_pairs_device(p0, p1, B, H, W, flow.data_ptr())
    torch.cuda.synchronize()
    L = _lib.load()
    n = L.tf_dbg_launch_profile(eng._h, None, None, None, None, 0)
    lv, wp, it = (np.zeros(n, np.int32) for _ in range(3))
    ms = np.zeros(n, np.float32)
    ptr = lambda x: x.ctypes.data_as(C.c_void_p)
    L.tf_dbg_launch_profile(eng._h, ptr(lv), ptr(wp), ptr(it), ptr(ms), n)
    iters = eng.last_iters()                    # [B, levels, warps, 2] (inner, outer)
    inner = iters[..., 0]
    nl = inner.shape[1]
    # level index in the records: pyramid level (nl-1 = coarsest ... 0 = full size)
    sizes = {}
    s = float(H)
    hh, ww = H, W
    px = []
    for l in range(nl):
        px.append(hh * ww)
        hh, ww = int(round(hh * 0.8)), int(round(ww * 0.8))
    active = np.array([(inner[:, nl - 1 - l if False else l, w] > i).sum() for l, w, i in zip(lv, wp, it)])
    act2 = np.array([np.minimum(np.maximum(inner[:, l, w] - i, 0), a.k).sum() for l, w, i in zip(lv, wp, it)])   # pair-iterations
    work = act2 * np.array([px[l] for l in lv], dtype=np.float64)                  # px-iterations in the launch
    print(f"B={B} {H}x{W}: {n} launches, {ms.sum():.1f} ms in tvl1_iter, device total {st['ms_device']:.1f} ms")
    full = work / ms
    peak = np.percentile(full[active == B], 90) if np.any(active == B) else full.max()
    print(f"rate at full activity (p90): {peak / 1e6:.1f} Mpx-it/ms;  ideal time at that rate {work.sum() / peak:.1f} ms "
          f"=> lock-step/tail efficiency {work.sum() / peak / ms.sum():.2f}")
    print("level  launches   ms    work-share  efficiency-vs-peak")
    for l in sorted(set(lv)):
        m = lv == l
        print(f"  {l}    {m.sum():5d}  {ms[m].sum():7.2f}   {work[m].sum() / work.sum():6.3f}     {work[m].sum() / peak / ms[m].sum():5.2f}")
    print("active-pair bucket  launches   ms     efficiency-vs-peak")
    edges = [0, 1, 2, 4, 8, 16, 32, B - 1, B]
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (active > lo) & (active <= hi)
        if m.any():
            print(f"  ({lo:3d},{hi:3d}]          {m.sum():5d}  {ms[m].sum():7.2f}   {work[m].sum() / peak / max(ms[m].sum(), 1e-9):5.2f}")
    m = active == 0
    print(f"  none active         {m.sum():5d}  {ms[m].sum():7.2f}")


if __name__ == "__main__":
    main()
