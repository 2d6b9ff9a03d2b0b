#!/bin/bash
# split of the 8 fresh conv1-rebuild blocks over the four consumer waves of block 0's weight gradient (SED_DBG bits 12-13)
for dbg in 0 4096 8192 12288; do
  echo "=== SED_DBG=$dbg"
  SED_DBG=$dbg python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > /tmp/c1s.json
  python tools/show_bench.py /tmp/c1s.json | sed -n '1p;4,5p'
done
