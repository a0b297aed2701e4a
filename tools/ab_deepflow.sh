#!/bin/bash
# A/B of DeepFlow engine knobs on the GPU box: bash tools/ab_deepflow.sh "sor_rt=1,sor_fuse=5" "sor_rt=0" ...
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
for cfg in "$@"; do
  timeout -k 10 300 python3 bench.py --algo deepflow --batch ${DF_BATCH:-128} --steps 3 --no-cpu-baseline --tuning "$cfg" > gpurun_out/ab/df.json 2> gpurun_out/ab/df.err && python3 - "$cfg" <<'PY' || { echo "FAILED $cfg"; tail -3 gpurun_out/ab/df.err; }
import json, sys
d = json.load(open("gpurun_out/ab/df.json")); r = d["roofline"]
print("%-36s pairs/s %6.1f ms/step %7.2f | SOR launches/step %5.0f avg ms %.4f px-sweeps/s %.1fG latency %.1f ms" % (
    sys.argv[1], d["value"], d["ms_per_step"], r["launches_per_step"], r["avg_launch_ms"] or 0, d["px_sweeps_per_s"] / 1e9, d.get("latency_ms_single_pair", 0)))
PY
done
