#!/bin/bash
# do the layer kernels run faster per clip when the whole working set of a launch fits the 256 MB Infinity Cache? (GPU box)
for shape in "32 3000 32 64 64" "8 3000 32 64 64" "4 3000 32 64 64" "32 6001 64 32 32" "8 6001 64 32 32"; do
  timeout -k 10 120 python tools/bench_layer.py $shape 20 2>&1 | grep -E "layer|wgrad  PRO_BNRELU DZ_POOL \(\+|fwd    PRO_BNRELU EPI_STATS|dgrad  PRO_NONE   EPI_RELU|dgrad  PRO_NONE   EPI_STORE" || exit 1
done
