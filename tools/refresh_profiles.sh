#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace summaries of the bench commands and the two PMC passes
# behind profiles/hbm_traffic.json.  Outputs under gpurun_out/prof/; tools/collect_profiles.py copies the summaries
# into profiles/.   usage: bash tools/refresh_profiles.sh <round-tag, e.g. r01>
set -e
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run_stats() {   # name, bench args...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python3 bench.py "$@" > $OUT/${TAG}_bench_$name.json 2> $OUT/$name.err
  echo "stats $name done"
}
run_stats default
run_stats lanes1 --lanes 1
run_stats deepflow_lanes1 --algo deepflow --batch 64 --steps 2 --lanes 1 --cpu-sample 2
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-include-regex k_iter2 --output-format csv -d $OUT/pmc_$c -o pmc_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --lanes 1 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
  echo "pmc $c done"
done
find $OUT -name "*.csv" | head -40
