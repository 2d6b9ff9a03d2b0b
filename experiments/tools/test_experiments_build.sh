#!/bin/bash
# Builds the library with make EXPERIMENTS=1 (the measured-slower kernels: resident-weight convs, cin-sliced fused backward, dz-producing
# data gradient, tap-split block-0 backward ...) on the GPU box, runs the tests that need it; the product build is restored (and compared)
# by the EXIT trap of tools/lib_restore.sh.
set -e
mkdir -p gpurun_out
source tools/lib_restore.sh
cd soundeventdetection-pytorch_amd/csrc
rm -f *.o
make -j14 EXPERIMENTS=1 > /tmp/mk_exp.log 2>&1 || (tail -20 /tmp/mk_exp.log; exit 1)
cd ../..
python -m pytest experiments/tests -m gpu -q 2>&1 | tail -4
python -m pytest tests/test_gpu_kernels_oracle.py tests/test_gpu_kernels_ab.py -m gpu -q 2>&1 | tail -4
python experiments/tools/ab_dgrad_dz.py 2 2>&1 | grep -E "bit-identical|sum of medians: w"
