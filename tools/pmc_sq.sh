#!/bin/bash
# Runs on the GPU box (through gpurun): SQ/GRBM counter passes over the dominant kernels of the bench workload, a few
# counters per pass (8 SQ slots, MI355X_MICROARCH.md "rocprofv3 PMC slots"), --pmc alone (never with a trace option).
# Raw CSVs land under gpurun_out/pmc_sq_<tag>/; tools/collect_pmc.py condenses them into profiles/.
# usage: bash tools/pmc_sq.sh <round-tag> [TVL1|deepflow]
set -u
TAG=${1:-r03}
ALGO=${2:-TVL1}
KREGEX=${KREGEX:-"k_iter2_rows|k_df_sor_rt"}     # other kernels: KREGEX="k_median2|k_warp" bash tools/pmc_sq.sh r02x
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_${TAG}_$ALGO
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
[ -f $OUT/../counters_list.txt ] || rocprofv3 -L > $OUT/../counters_list.txt 2>&1
EXTRA="--batch 128"          # one sub-batch on one lane: exclusive launches (the default 384-pair step runs three lanes at once)
[ -n "${TUNING:-}" ] && EXTRA="$EXTRA --tuning $TUNING"          # engine knobs for an experiment: TUNING=sor_fuse=7 bash tools/pmc_sq.sh x deepflow
pass() {
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-include-regex "$KREGEX" --output-format csv -d $OUT/$name -o $name -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --steps-only --in-flight 1 --lanes 1 --no-deepflow --algo $ALGO $EXTRA \
    > $OUT/$name.json 2> $OUT/$name.err
  echo "pass $name ($*) rc=$?"
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
pass sq3 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass sq4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq5 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
find $OUT -name "*counter_collection.csv" | head
