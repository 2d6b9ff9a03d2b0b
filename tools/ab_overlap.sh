#!/bin/bash
# A/B of the front-end overlap modes (bench.py --overlap-frontend 0..3), interleaved twice on one box.
out=gpurun_out/r04_overlap; mkdir -p $out
for rep in 1 2 3; do
  for m in 0 1 2 3; do
    python3 bench.py --overlap-frontend $m --steps 300 --warmup 30 --no-cpu-baseline > $out/m${m}_r${rep}.json 2> $out/m${m}_r${rep}.err || { echo "mode $m failed"; tail -5 $out/m${m}_r${rep}.err; }
    python3 - <<PY
import json
try:
    d=json.load(open("$out/m${m}_r${rep}.json")); print("mode $m rep $rep: %.3f ms/step  %.0f clips/s  sum_of_kernels %.3f" % (d["ms_per_step"], d["value"], d["gpu_time_ms_per_step_sum_of_kernels"]))
except Exception as e: print("mode $m rep $rep: no result", e)
PY
  done
done
