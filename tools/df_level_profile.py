#!/usr/bin/env python3
"""Per-level, per-kernel device time of a DeepFlow solve from a rocprofv3 --kernel-trace CSV.
usage: python tools/df_level_profile.py <dir with *_kernel_trace.csv> [pairs]
Launches are attributed to pyramid levels by their order (the solve walks the levels coarse -> fine); the SOR launches
are additionally listed by grid size (= level geometry) with their mean duration."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not files:
        sys.exit("no kernel_trace.csv under " + d)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by_kernel = defaultdict(float)
    by_grid = defaultdict(lambda: [0, 0.0])
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        by_kernel[name] += dur
        if "sor" in name:
            key = (name[-40:], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
            by_grid[key][0] += 1
            by_grid[key][1] += dur
    tot = sum(by_kernel.values())
    print(f"total kernel time {tot / 1e3:.2f} ms over {len(rows)} launches")
    for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1])[:14]:
        print(f"  {v / 1e3:9.3f} ms  {100 * v / tot:5.1f} %  {k[-70:]}")
    print("SOR launches by grid (tiles x, tiles y, pairs): count, mean us, total ms")
    for k, (n, t) in sorted(by_grid.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {k[0]:>40s} {k[1]:3d} x {k[2]:3d} x {k[3]:3d}: {n:5d}  {t / n:9.1f} us  {t / 1e3:8.3f} ms")


if __name__ == "__main__":
    main()
