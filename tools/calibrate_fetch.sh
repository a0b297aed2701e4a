#!/bin/bash
# On the GPU box: FETCH_SIZE / WRITE_SIZE of streaming copies with known byte counts -> gpurun_out/<tag>_fetch_calibration.json
# (copy it to profiles/).  One counter per pass, --pmc alone.   usage: bash tools/calibrate_fetch.sh r03
set -u
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/fetch_calib_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o cal -- $GRAFT_REPO_ROOT/tools/microbench/fetch_calib > $OUT/$c.log 2>&1
  echo "pass $c rc=$?"
done
python3 - $OUT $TAG <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
B = float(1 << 30)
res = {"what": "rocprofv3 FETCH_SIZE / WRITE_SIZE (KB) of streaming copies that read and write exactly 1 GiB each; factor = true bytes / (counter x 1024)",
       "command": "bash tools/calibrate_fetch.sh " + tag + "  (tools/microbench/fetch_calib under rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE)", "kernels": {}}
names = {"k_copy<float>": "4", "k_copy<HIP_vector_type<float, 2u>": "8", "k_copy<HIP_vector_type<float, 4u>": "16", "k_copy_tile8": "8_tile_rows"}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/" + ctr + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            for k, key in names.items():
                if k in r["Kernel_Name"]:
                    d = res["kernels"].setdefault(key, {})
                    d[ctr + "_kb"] = d.get(ctr + "_kb", 0.0) + float(r["Counter_Value"])
for key, d in res["kernels"].items():
    if "FETCH_SIZE_kb" in d: d["fetch_factor"] = B / (d["FETCH_SIZE_kb"] * 1024.0)
    if "WRITE_SIZE_kb" in d: d["write_factor"] = B / (d["WRITE_SIZE_kb"] * 1024.0)
json.dump(res, open(out + "/../" + tag + "_fetch_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
