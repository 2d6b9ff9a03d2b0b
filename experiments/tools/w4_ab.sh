#!/bin/bash
# parity of the one-wave-per-SIMD kernel, then layer timings against the producer/consumer kernel (GPU box)
timeout -k 10 300 python -m pytest tests/test_gpu_kernels_ab.py -x -q -k "one_wave_per_simd" 2>&1 | tail -5 || exit 1
for k in ${1:-p 4}; do
  echo "=== SED_CONV_KERNEL=$k"
  for shape in "32 3000 32 64 64" "32 1500 16 64 128" "32 1500 16 128 128" "32 750 8 128 128"; do
    SED_CONV_KERNEL=$k timeout -k 10 120 python tools/bench_layer.py $shape 20 2>&1 | grep -E "layer|fwd|dgrad" || exit 1
  done
done
