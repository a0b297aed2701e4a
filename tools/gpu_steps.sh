#!/bin/bash
# Run GPU steps one after another; a step that times out or is killed ends the whole call (no further GPU step after a hang),
# an ordinary failure (assertion, non-zero exit) is recorded and the next step still runs.
# usage: bash tools/gpu_steps.sh <tag> "<cmd1>" "<cmd2>" ...   (logs: gpurun_out/<tag>_<k>.log)
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out
k=0
for cmd in "$@"; do
  k=$((k+1))
  echo "== step $k: $cmd" | tee gpurun_out/${tag}_$k.log
  timeout -k 10 ${STEP_TIMEOUT:-900} bash -c "$cmd" >> gpurun_out/${tag}_$k.log 2>&1
  rc=$?
  echo "== step $k rc=$rc"
  tail -n ${TAIL:-6} gpurun_out/${tag}_$k.log
  if [ $rc -ge 124 ]; then echo "step $k timed out / was killed: stopping"; exit $rc; fi
done
exit 0
