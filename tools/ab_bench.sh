#!/bin/bash
# usage: ab.sh "<tuning> <extra bench args>" ...   (runs on the GPU box from the repo root)
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
for cfg in "$@"; do
  set -- $cfg
  t=$1; shift
  timeout -k 10 200 python3 bench.py --no-deepflow --no-cpu-baseline --steps 5 --tuning $t "$@" > gpurun_out/ab/ab.json 2> gpurun_out/ab/ab.err && python3 - "$cfg" <<'PY' || { echo "FAILED $cfg"; tail -3 gpurun_out/ab/ab.err; }
import json, sys
d = json.load(open("gpurun_out/ab/ab.json")); r = d["roofline"]; v = r.get("valu", {})
print("%-40s pairs/s %6.0f ms/step %6.2f px-it/s %6.1fG | iter launches/step %5.0f avg ms %.4f rate all %6.1fG full %s parity-n/a" % (
    sys.argv[1], d["value"], d["ms_per_step"], d["px_iterations_per_s"] / 1e9, r["launches_per_step"], r["avg_launch_ms"] or 0,
    (v.get("px_iterations_per_s_all_launches") or 0) / 1e9, "%.1fG" % (v["px_iterations_per_s_full_launches"] / 1e9) if v.get("px_iterations_per_s_full_launches") else "-"), d.get("stage_ms_per_step"))
PY
done
