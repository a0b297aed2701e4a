#!/bin/bash
# Runs on the GPU box (through gpurun): everything profiles/ is built from, for the CURRENT kernel build.
#   1. python3 bench.py --pmc : the FETCH_SIZE / WRITE_SIZE passes (children of bench.py, one counter per pass) for the
#      DualTVL1 and the DeepFlow leg, and the JSON line that carries the traffic measured that way
#   2. SQ / GRBM counter passes over the dominant kernels (tools/pmc_sq.sh)
#   3. rocprofv3 --kernel-trace --stats summaries of the default bench command and of its single-lane form (they read 1 and 2)
#   4. a kernel trace of the timed steps (tools/timed_regime.py), the saliency preprocessing alone, the queue forms side by side
#   5. per-launch profile of the lock-step driver (tools/launch_profile.py) and the arithmetic ablation probe
# Outputs under gpurun_out/prof_<tag>/ ; tools/collect_profiles.py <tag> condenses them into profiles/.
# usage: bash tools/refresh_profiles.sh <round-tag, e.g. r03>
set -u
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run_stats() {   # name, bench args...
  local name=$1; shift
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/${TAG}_bench_$name.json 2> $OUT/$name.err
  echo "stats $name rc=$?"
}
# counters first: the pass measures the traffic of THIS build; copied into profiles/ so that the stats runs below carry it
timeout -k 10 900 python3 $GRAFT_REPO_ROOT/bench.py --pmc --pmc-dir $OUT/pmc_live --round-tag $TAG > $OUT/${TAG}_bench_pmc.json 2> $OUT/bench_pmc.err
echo "bench --pmc rc=$?"
[ -f $OUT/pmc_live/hbm_traffic.json ] && cp $OUT/pmc_live/hbm_traffic.json $GRAFT_REPO_ROOT/profiles/hbm_traffic.json
# SQ / GRBM counters next, condensed into profiles/ right away: bench.py derives roofline.limiter from the record of THIS build
cd $GRAFT_REPO_ROOT
bash tools/pmc_sq.sh $TAG TVL1 > $OUT/pmc_sq_tvl1.log 2>&1; tail -2 $OUT/pmc_sq_tvl1.log
bash tools/pmc_sq.sh $TAG deepflow > $OUT/pmc_sq_df.log 2>&1; tail -2 $OUT/pmc_sq_df.log
python3 tools/collect_profiles.py $TAG $GRAFT_REPO_ROOT/gpurun_out/collected_$TAG > $OUT/collect0.log 2>&1
[ -f gpurun_out/collected_$TAG/${TAG}_sq_counters.json ] && cp gpurun_out/collected_$TAG/${TAG}_sq_counters.json profiles/
cd /tmp
# the regime `value` is measured in: a kernel trace of the timed steps alone (three lanes' kernels sharing the GPU) -> <tag>_timed_regime.json
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/regime -o regime -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --steps-only --no-deepflow --no-cpu-baseline --no-profile > $OUT/regime.json 2> $OUT/regime.err
echo "regime trace rc=$?"
python3 $GRAFT_REPO_ROOT/tools/timed_regime.py $OUT/regime $TAG --steps 6 --out $GRAFT_REPO_ROOT/gpurun_out/collected_$TAG > $OUT/timed_regime.log 2>&1; tail -1 $OUT/timed_regime.log
[ -f $GRAFT_REPO_ROOT/gpurun_out/collected_$TAG/${TAG}_timed_regime.json ] && cp $GRAFT_REPO_ROOT/gpurun_out/collected_$TAG/${TAG}_timed_regime.json $GRAFT_REPO_ROOT/profiles/
run_stats default
run_stats lanes1 --batch 128 --lanes 1 --no-cpu-baseline
# one algorithm per profiled command (the combined tables need arithmetic to read per-algorithm shares)
run_stats tvl1_lanes1 --batch 128 --lanes 1 --no-cpu-baseline --no-deepflow
run_stats deepflow_lanes1 --lanes 1 --no-cpu-baseline --algo deepflow
# the no_saliency=False preprocessing by itself
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/saliency -o saliency -- python3 $GRAFT_REPO_ROOT/tools/saliency_bench.py > $OUT/${TAG}_saliency_bench.txt 2> $OUT/saliency.err
echo "saliency rc=$?"
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/queue_forms.py --steps 24 --rounds 2 --distinct > $OUT/${TAG}_queue_forms.txt 2>&1; echo "queue_forms rc=$?"
timeout -k 10 200 python3 tools/launch_profile.py --batch 64 > $OUT/${TAG}_launch_profile_b64.txt 2>&1; echo "launch_profile rc=$?"
[ -x tools/microbench/ablate_probe ] && (cd tools/microbench && timeout -k 10 200 ./ablate_probe > $OUT/${TAG}_ablate_probe.txt 2>&1; echo "ablate rc=$?")
# condense on the box (the raw kernel traces and counter CSVs are tens of MB; gpurun carries at most 64 MiB back) and drop the raw files
python3 tools/collect_profiles.py $TAG $GRAFT_REPO_ROOT/gpurun_out/collected_$TAG > $OUT/collect.log 2>&1; tail -3 $OUT/collect.log
rm -rf $OUT/default $OUT/lanes1 $OUT/tvl1_lanes1 $OUT/deepflow_lanes1 $OUT/regime $OUT/saliency $OUT/pmc_live/pmc_live_* $GRAFT_REPO_ROOT/gpurun_out/pmc_sq_${TAG}_TVL1/sq? $GRAFT_REPO_ROOT/gpurun_out/pmc_sq_${TAG}_deepflow/sq?
du -sh $GRAFT_REPO_ROOT/gpurun_out
