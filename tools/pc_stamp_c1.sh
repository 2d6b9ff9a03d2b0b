#!/bin/bash
# phase stamps of block 0's C1-mode forward (make STAMPS=1 DEBUG_SWITCHES=1, in-tree; SED_DBG=16 prints them); the product build is
# restored by the EXIT trap of tools/lib_restore.sh
set -e
mkdir -p gpurun_out/r04g
source tools/lib_restore.sh
cd soundeventdetection-pytorch_amd/csrc
rm -f *.o
make -j14 STAMPS=1 DEBUG_SWITCHES=1 "$@" > /tmp/mk_st.log 2>&1 || (tail -20 /tmp/mk_st.log; exit 1)
cd ../..
SED_DBG=16 timeout -k 10 200 python tools/pc_stamp.py c1 2>&1 | grep -v amdgpu | tail -8
