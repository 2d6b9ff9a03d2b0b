#!/bin/bash
# Interleaved A/B of the bench step between two values of ONE environment knob, in alternating fresh processes on one box:
#   tools/ab_env.sh SED_WGRAD_REDUCE inline batch [rounds] [extra bench.py args...]
# prints ms/step of every run and the medians (the boxes of the pool differ by 5-15 %: only same-box alternations compare).
var=$1; a=$2; b=$3; rounds=${4:-3}; shift 4 2>/dev/null
out=$(mktemp -d)
for r in $(seq 1 $rounds); do
  for v in $a $b; do
    env $var=$v python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-measured-peaks "$@" 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['ms_per_step'],4), 'ms/step', 'sum of kernels', round(d['gpu_time_ms_per_step_sum_of_kernels'],4))" | tee -a $out/$v.txt
  done
done
python3 - <<PY
import statistics
for v in ("$a", "$b"):
    xs = [float(l.split()[1]) for l in open("$out/" + v + ".txt")]
    print("$var=" + v, "median", statistics.median(xs), "ms/step over", len(xs), "runs")
PY
