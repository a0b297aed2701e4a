#!/bin/bash
# kernel-trace of one DeepFlow solve per batch size: bash tools/df_prof.sh <tag> "<DF_TUNING>" B1 B2 ...
export TMPDIR=/tmp
tag=$1; tun=$2; shift 2
for B in "$@"; do
  out=gpurun_out/$tag/b$B
  mkdir -p $out
  DF_TUNING="$tun" rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/df_bench.py $B 5 > $out/run.log 2>&1
  tail -1 $out/run.log
  python3 tools/df_level_profile.py $out $B > $out/levels.txt 2>&1
  head -60 $out/levels.txt
done
