#!/usr/bin/env python3
"""Copy what tools/refresh_profiles.sh left under gpurun_out/prof/ into profiles/ (kernel_stats summaries, the JSON
lines the profiled runs printed) and recompute profiles/hbm_traffic.json from the two PMC passes.
usage: python tools/collect_profiles.py r01"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def counter_mean(name):
    files = glob.glob(os.path.join(SRC, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)
    vals = {}
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == name and "k_iter2_rows" in row["Kernel_Name"]:
                    vals.setdefault(row["Dispatch_Id"], 0.0)
                    vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
    v = list(vals.values())
    return (sum(v) / len(v), len(v), max(v)) if v else (None, 0, None)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    for name in ("default", "lanes1", "deepflow_lanes1"):
        st = glob.glob(os.path.join(SRC, name, "**", "*kernel_stats.csv"), recursive=True)
        if st:
            shutil.copy(st[0], os.path.join(DST, f"{tag}_bench_{name}_kernel_stats.csv"))
        js = os.path.join(SRC, f"{tag}_bench_{name}.json")
        if os.path.exists(js) and os.path.getsize(js):
            shutil.copy(js, os.path.join(DST, f"{tag}_bench_{name}.json"))
    f, nf, fmax = counter_mean("FETCH_SIZE")
    w, nw, wmax = counter_mean("WRITE_SIZE")
    if f is not None and w is not None:
        out = {
            "tvl1_iter_bytes_per_launch": (2 * f + w) * 1024,
            "kernel": "k_iter2_rows",
            "launches_profiled": nf,
            "fetch_size_kb_mean": f,
            "write_size_kb_mean": w,
            "full_level0_launch_bytes": (2 * fmax + wmax) * 1024,
            "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies "
                       "128-B read requests at 64 B); separate --pmc passes",
            "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-include-regex k_iter2 -- python3 bench.py --steps 1 --warmup 0 "
                       "--no-cpu-baseline --no-profile --lanes 1",
            "round": int(tag[1:]) if tag[1:].isdigit() else tag,
        }
        with open(os.path.join(DST, "hbm_traffic.json"), "w") as fh:
            json.dump(out, fh, indent=1)
        print(json.dumps(out, indent=1))
    for p in sorted(os.listdir(DST)):
        print(p)


if __name__ == "__main__":
    main()
