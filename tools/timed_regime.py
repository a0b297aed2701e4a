#!/usr/bin/env python3
"""roofline.timed_regime: the dominant kernel in the regime bench.py's `value` is measured in -- several lanes'
kernels sharing the GPU -- from ONE rocprofv3 --kernel-trace of the timed steps.  bench.py's own roofline figures come from
instrumented single-lane repeats (a launch to itself); here, for the same kernel:
  * mean duration of a launch while it shares the GPU, and the mean number of kernels in flight while it runs;
  * its share-adjusted duration: the integral over the launch of 1 / (kernels running), so that the share-adjusted durations of all
    kernels add up to the time the GPU was busy -- bytes per launch / that = the bandwidth the kernel is worth in this regime.
Reads the trace of
    rocprofv3 --kernel-trace --output-format csv -d DIR -o trace -- python3 bench.py --steps K --warmup 1 --steps-only --no-deepflow --no-cpu-baseline --no-profile
(the window analysed is the last K/(K+2) of the dominant kernel's launches: set-up call and warm-up step dropped) and writes
profiles/<tag>_timed_regime.json, which bench.py attaches to its line when the kernel-source fingerprint matches.
usage: python3 tools/timed_regime.py DIR TAG [--steps K] [--kernel k_iter2_rows] [--out profiles]"""
import argparse
import csv
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("tag")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--kernel", default="k_iter2_rows")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles"))
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    rows = []
    for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    if not rows:
        raise SystemExit(f"no *kernel_trace.csv under {a.dir}")
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
    t1 = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
    dom = np.array([a.kernel in r["Kernel_Name"] for r in rows])
    idx = np.flatnonzero(dom)
    if len(idx) < 10:
        raise SystemExit(f"only {len(idx)} launches of {a.kernel} in the trace")
    # the timed steps: the last K of K + 2 equal batches (set-up call + one warm-up step precede them)
    first = idx[len(idx) * 2 // (a.steps + 2)]
    w0, w1 = t0[first], t1[idx[-1]]
    keep = (t1 > w0) & (t0 < w1)
    s, e, d = np.clip(t0[keep], w0, w1), np.clip(t1[keep], w0, w1), dom[keep]
    # sweep: n(t) = kernels running; per launch the integral of 1/n over its interval
    ev = sorted([(int(x), 1, i) for i, x in enumerate(s)] + [(int(x), -1, i) for i, x in enumerate(e)])
    running, last = set(), ev[0][0]
    adj = np.zeros(len(s))
    conc_time = np.zeros(len(s))             # integral of n(t) over each launch's interval
    busy = 0
    hist = {}
    for t, kind, i in ev:
        if t > last and running:
            n = len(running)
            dt = t - last
            busy += dt
            hist[min(n, 6)] = hist.get(min(n, 6), 0) + dt
            for j in running:
                adj[j] += dt / n
                conc_time[j] += dt * n
        last = t
        if kind == 1:
            running.add(i)
        else:
            running.discard(i)
    dur = (e - s).astype(np.float64)
    dd = dur[d]
    out = {
        "kernel": a.kernel, "round": a.tag, "launches_in_window": int(d.sum()), "window_ms": (w1 - w0) / 1e6, "gpu_busy_ms": busy / 1e6,
        "mean_duration_us": float(dd.mean() / 1e3), "mean_kernels_in_flight_while_it_runs": float((conc_time[d] / np.maximum(dur[d], 1)).mean()),
        "mean_share_adjusted_duration_us": float(adj[d].mean() / 1e3),
        "share_of_gpu_busy_time": float(adj[d].sum() / busy),
        "kernels_in_flight_share_of_window": {str(k) + ("+" if k == 6 else ""): v / (w1 - w0) for k, v in sorted(hist.items())},
        "idle_share_of_window": 1.0 - busy / (w1 - w0),
        "command": a.command or "rocprofv3 --kernel-trace -- python3 bench.py --steps K --warmup 1 --steps-only --no-deepflow --no-cpu-baseline --no-profile",
        "note": "share-adjusted duration = integral over the launch of 1 / (kernels running): the launch's part of the GPU's busy time when several lanes' kernels overlap; "
                "bytes per launch / this = the bandwidth the kernel is worth in the regime `value` is measured in",
    }
    from bench import kernel_source_fingerprint
    out["source_fingerprint"] = kernel_source_fingerprint()
    os.makedirs(a.out, exist_ok=True)
    path = os.path.join(a.out, f"{a.tag}_timed_regime.json")
    with open(path, "w") as f:
        json.dump({a.kernel: out}, f, indent=1)
    print(json.dumps(out, indent=1))
    print("wrote", path)


if __name__ == "__main__":
    main()
