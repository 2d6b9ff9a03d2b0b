#!/bin/bash
# Interleaved A/B of the bench step between two BUILDS of libsed_hip.so, in alternating fresh processes on one box (GPU box: the tree there
# is a scratch copy).  The two libraries are built beforehand into soundeventdetection-pytorch_amd/ab/lib_a.so (before) and lib_b.so (after):
#   git stash; make -C soundeventdetection-pytorch_amd/csrc; cp .../libsed_hip.so .../ab/lib_a.so; git stash pop; make ...; cp ... ab/lib_b.so
#   tools/ab_lib.sh [rounds] [extra bench.py args...]
rounds=${1:-3}; shift 1 2>/dev/null
pk=soundeventdetection-pytorch_amd
out=$(mktemp -d)
for r in $(seq 1 $rounds); do
  for v in a b; do
    cp $pk/ab/lib_$v.so $pk/libsed_hip.so
    python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-measured-peaks "$@" 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib_$v', round(d['ms_per_step'],4), 'ms/step', 'sum of kernels', round(d['gpu_time_ms_per_step_sum_of_kernels'],4))" | tee -a $out/$v.txt
  done
done
cp $pk/ab/lib_b.so $pk/libsed_hip.so
python3 - <<PY
import statistics
for v in ("a", "b"):
    xs = [float(l.split()[1]) for l in open("$out/" + v + ".txt")]
    print("lib_" + v, "median", statistics.median(xs), "ms/step over", len(xs), "runs")
PY
