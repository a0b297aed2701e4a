#!/bin/bash
# A/B of two library builds, interleaved: bash tools/ab_lib.sh <libA> <libB> [rounds]
export TMPDIR=/tmp
A=$1; B=$2; N=${3:-3}
for r in $(seq 1 $N); do
  for L in $A $B; do
    TEEFLOW_LIB=$PWD/$L timeout -k 10 200 python3 bench.py --no-deepflow --no-cpu-baseline --steps 8 --steps-only > gpurun_out/ab_lib.json 2> gpurun_out/ab_lib.err && python3 - $L <<'PY' || { echo "FAILED $L"; tail -3 gpurun_out/ab_lib.err; }
import json, sys
d = json.load(open("gpurun_out/ab_lib.json")); r = d["roofline"]
s=d["stage_ms_per_step"]; print("%-32s pairs/s %6.0f ms/step %6.2f | iter %6.2f warp %5.2f median %5.2f avg launch %.4f Gpx-it/s %.1f" % (sys.argv[1], d["value"], d["ms_per_step"], s["tvl1_iter"], s["warp"], s["median"], r["avg_launch_ms"], d.get("px_iterations_per_s", 0) / 1e9))
PY
  done
done
