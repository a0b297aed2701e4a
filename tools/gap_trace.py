#!/usr/bin/env python3
"""Where a solve's wall time goes that is not kernel time: from a rocprofv3 --kernel-trace CSV (last solve only = from the last
k_u8_to_f32 on), per kernel name: launches, kernel time, and the idle gap BEFORE each launch of that name (end of the previous
kernel on the device to this one's start), so that a launch path that costs more than the others stands out.
usage: python tools/gap_trace.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

import numpy as np


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_u8_to_f32" in r["Kernel_Name"]]
    rows = rows[starts[-1]:] if starts else rows
    t = np.array([[int(r["Start_Timestamp"]), int(r["End_Timestamp"])] for r in rows], dtype=np.int64)
    name = [r["Kernel_Name"].split("(")[0].replace("void ", "") for r in rows]
    dur = (t[:, 1] - t[:, 0]) / 1e3
    gap = np.concatenate([[0.0], np.maximum(t[1:, 0] - np.maximum.accumulate(t[:-1, 1]), 0) / 1e3])
    span = (t[-1, 1] - t[0, 0]) / 1e3
    print(f"{len(rows)} launches, span {span / 1e3:.2f} ms, kernel time {dur.sum() / 1e3:.2f} ms, idle gaps {gap.sum() / 1e3:.2f} ms "
          f"(median {np.median(gap):.2f} us, mean {gap.mean():.2f} us, p90 {np.percentile(gap, 90):.2f} us)")
    by = defaultdict(lambda: [0, 0.0, []])
    for n, x, g in zip(name, dur, gap):
        by[n][0] += 1; by[n][1] += x; by[n][2].append(g)
    print("kernel ms   launches  mean us | gap before it: total ms  median us  mean us   name")
    for n, (c, x, g) in sorted(by.items(), key=lambda kv: -(kv[1][1] + sum(kv[1][2])))[:14]:
        g = np.array(g)
        print(f"  {x / 1e3:8.3f} {c:7d} {x / c:9.2f} | {g.sum() / 1e3:8.3f} {np.median(g):9.2f} {g.mean():9.2f}   {n[-48:]}")
    # by image width (grid size tells the level): gaps and kernel time per decile of the launch sequence
    k = len(rows) // 10
    print("tenths of the launch sequence (coarse -> fine): kernel ms, gap ms")
    for i in range(10):
        s = slice(i * k, (i + 1) * k if i < 9 else len(rows))
        print(f"  {i}: {dur[s].sum() / 1e3:8.2f} {gap[s].sum() / 1e3:8.2f}   mean kernel {dur[s].mean():7.1f} us, mean gap {gap[s].mean():6.2f} us")


if __name__ == "__main__":
    main()
