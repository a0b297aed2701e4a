#!/bin/bash
# A/B of engine knobs through bench.py's timed region on ONE box, interleaved rounds: usage: bash tools/bench_tunings.sh ROUNDS "<bench args>" cfg1 cfg2 ...   (cfg = tuning string, "-" = defaults)
rounds=$1; shift
args=$1; shift
for r in $(seq 1 $rounds); do
  for cfg in "$@"; do
    t=""; [ "$cfg" != "-" ] && t="--tuning $cfg"
    v=$(timeout -k 10 300 python bench.py --steps 20 --warmup 5 --steps-only --no-profile --no-cpu-baseline --no-deepflow $args $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f pairs/s %.2f ms' % (d['value'], d['ms_per_step']))") || exit 124
    echo "round $r  $cfg : $v"
  done
done
