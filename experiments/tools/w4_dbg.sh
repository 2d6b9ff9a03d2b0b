#!/bin/bash
for dbg in 0 1 8 9; do
  echo "=== SED_DBG=$dbg"
  SED_DBG=$dbg SED_CONV_KERNEL=4 timeout -k 10 120 python tools/bench_layer.py 32 1500 16 128 128 20 2>&1 | grep -E "fwd|dgrad" || exit 1
done
