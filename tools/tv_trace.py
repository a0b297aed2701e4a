#!/usr/bin/env python3
"""What a tvl1_iter launch costs beside its bytes: from a rocprofv3 --kernel-trace CSV of tools/tv_trace_run.py (second solve only),
per kernel: launches, total time, and for the iteration kernel the duration of launches by grid size, the idle gaps between
consecutive kernels of the stream, and a fit  duration = T0 + bytes / rate  over the launches with every pair active.
usage: python tools/tv_trace.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

import numpy as np


def main():
    d = sys.argv[1]
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the second solve starts at the second k_u8_to_f32 / k_f32_to_level0
    starts = [i for i, r in enumerate(rows) if "k_u8_to_f32" in r["Kernel_Name"]]
    rows = rows[starts[-1]:] if starts else rows
    t = np.array([[int(r["Start_Timestamp"]), int(r["End_Timestamp"])] for r in rows], dtype=np.int64)
    name = [r["Kernel_Name"].split("(")[0].replace("void ", "") for r in rows]
    dur = (t[:, 1] - t[:, 0]) / 1e3
    gap = np.maximum(t[1:, 0] - t[:-1, 1], 0) / 1e3
    span = (t[-1, 1] - t[0, 0]) / 1e3
    print(f"{len(rows)} launches, span {span / 1e3:.2f} ms, kernel time {dur.sum() / 1e3:.2f} ms, idle gaps {gap.sum() / 1e3:.2f} ms "
          f"(median gap {np.median(gap):.2f} us, mean {gap.mean():.2f} us)")
    by = defaultdict(lambda: [0, 0.0])
    for n, x in zip(name, dur):
        by[n][0] += 1; by[n][1] += x
    for n, (c, x) in sorted(by.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"  {x / 1e3:8.3f} ms {c:5d} launches  mean {x / c:8.2f} us  {n[-60:]}")
    it = [i for i, n in enumerate(name) if "k_iter2" in n]
    g = defaultdict(list)
    for i in it:
        g[(name[i][-24:], int(rows[i]["Grid_Size_X"]) // max(1, int(rows[i]["Workgroup_Size_X"])))].append(dur[i])
    print("iteration-kernel launches by (kernel, blocks): count, min / median / max us")
    for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:16]:
        v = np.array(v)
        print(f"  {k[0]:>24s} {k[1]:6d}: {len(v):4d}  {v.min():8.1f} {np.median(v):8.1f} {v.max():8.1f}   total {v.sum() / 1e3:7.3f} ms")


if __name__ == "__main__":
    main()


def concurrency(d):
    """two-lane runs: how much of the span has 0 / 1 / 2+ kernels in flight"""
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_u8_to_f32" in r["Kernel_Name"]]
    n_lanes = len(set(r.get("Queue_Id", "0") for r in rows))
    if len(starts) >= 2 * max(1, n_lanes // 1) and n_lanes > 1:
        rows = rows[starts[-n_lanes]:]
    ev = []
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
    ev.sort()
    lvl, last, acc = 0, ev[0][0], defaultdict(float)
    for t, dlt in ev:
        acc[min(lvl, 3)] += t - last
        last = t; lvl += dlt
    tot = sum(acc.values())
    print(f"queues {n_lanes}; span {tot / 1e6:.2f} ms; kernels in flight: " + ", ".join(f"{k}{'+' if k == 3 else ''}: {100 * v / tot:.1f} %" for k, v in sorted(acc.items())))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "conc":
    concurrency(sys.argv[1])
