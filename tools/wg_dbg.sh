#!/bin/bash
for dbg in 0 2; do
  echo "=== SED_DBG=$dbg"
  for shape in "32 3000 32 64 64" "32 1500 16 128 128"; do
    SED_DBG=$dbg timeout -k 10 120 python tools/bench_layer.py $shape 20 2>&1 | grep -E "layer|wgrad" || exit 1
  done
done
