#!/bin/bash
# Ablation of the producer/consumer weight-gradient kernel (GPU box): SED_DBG bits 1 no dz_out stores,
# 2 no consumer work, 8 no global loads, 16 MFMAs without LDS reads, 32 LDS reads without MFMAs.
for shape in "32 3000 32 64 64" "32 6001 64 32 32"; do
  for dbg in 0 2 8 10 16 32 24; do
    echo "=== shape $shape SED_DBG=$dbg"
    SED_DBG=$dbg python tools/bench_layer.py $shape 10 2>/dev/null | grep wgrad
  done
done
