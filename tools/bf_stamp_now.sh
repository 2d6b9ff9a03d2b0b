#!/bin/bash
# phase stamps of block 1's fused backward launches at the bench geometry (make STAMPS=1, in-tree); the product build is restored by the
# EXIT trap of tools/lib_restore.sh
set -e
source tools/lib_restore.sh
cd soundeventdetection-pytorch_amd/csrc
rm -f *.o
make -j14 STAMPS=1 "$@" > /tmp/mk_st.log 2>&1 || (tail -20 /tmp/mk_st.log; exit 1)
cd ../..
timeout -k 10 300 python tools/ab_fused.py 1 3 2>&1 | grep -E "bf producer|bf consumer|fused" | tail -12
