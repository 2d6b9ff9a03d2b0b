#!/bin/bash
# Ablation sweep of the conv-type kernels (GPU box): SED_DBG bits 1 = no output stores, 2 = no MFMA loop,
# 8 = no global loads.  usage: tools/ablate.sh > gpurun_out/ablate.log
set -e
for shape in "32 6001 64 32 32" "32 3000 32 64 64" "32 1500 16 128 128"; do
  for dbg in 0 1 2 8 3 10 11; do
    echo "=== shape $shape SED_DBG=$dbg"
    SED_DBG=$dbg python tools/bench_layer.py $shape 10
  done
done
