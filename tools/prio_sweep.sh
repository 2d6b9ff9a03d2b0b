#!/bin/bash
# A/B of the wave-priority switch (SED_DBG bits 8-9: loader waves, 10-11: MFMA waves) over the conv shapes (GPU box)
for dbg in 0 256 512 768 1024; do
  echo "=== SED_DBG=$dbg"
  for shape in "32 3000 32 64 64" "32 1500 16 128 128" "32 750 8 128 128"; do
    SED_DBG=$dbg python tools/bench_layer.py $shape 20 || exit 1
  done
  SED_DBG=$dbg python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STEP ms', d['ms_per_step'])" || exit 1
done
