#!/bin/bash
# as bench_tunings.sh, for the DeepFlow leg (128 pairs, 3 steps): bash tools/bench_tunings_df.sh ROUNDS cfg1 cfg2 ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for cfg in "$@"; do
    t=""; [ "$cfg" != "-" ] && t="--tuning $cfg"
    v=$(timeout -k 10 300 python bench.py --algo deepflow --steps 3 --warmup 1 --steps-only --no-profile --no-cpu-baseline $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f pairs/s %.2f ms' % (d['value'], d['ms_per_step']))") || exit 124
    echo "round $r  $cfg : $v"
  done
done
