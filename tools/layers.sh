#!/bin/bash
# bench_layer over the seven conv shapes of the BENCH workload (GPU box).  usage: tools/layers.sh [iters]
set -e
it=${1:-10}
for shape in "32 6001 64 32 32" "32 3000 32 32 64" "32 3000 32 64 64" "32 1500 16 64 128" "32 1500 16 128 128" "32 750 8 128 128"; do
  python tools/bench_layer.py $shape $it
done
