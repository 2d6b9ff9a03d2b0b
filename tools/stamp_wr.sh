#!/bin/bash
# WR path of the producer/consumer conv kernel on the GPU box: phase stamps (STAMPS + DEBUG_SWITCHES build), then the product build's A/B
set -e
cd soundeventdetection-pytorch_amd/csrc
make -j14 STAMPS=1 DEBUG_SWITCHES=1 > /tmp/mk.log 2>&1 || (tail -20 /tmp/mk.log; exit 1)
cd ../..
echo "== WR=1 128->128 fwd stats (pro=1)"; python tools/pc_stamp.py 32 1500 16 128 128 1 2>&1 | grep -v amdgpu.ids
cd soundeventdetection-pytorch_amd/csrc
make -j14 > /tmp/mk.log 2>&1 || (tail -20 /tmp/mk.log; exit 1)
cd ../..
python tools/ab_wr.py ${1:-5} 2>&1 | grep -v amdgpu.ids
