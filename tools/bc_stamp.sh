#!/bin/bash
# in-kernel phase stamps of the block-0 fused backward: rebuilds csrc/ IN-TREE with STAMPS=1; the product build is restored (and compared
# with the library found at start) by the EXIT trap of tools/lib_restore.sh, also after a failure or an interrupt
set -e
source tools/lib_restore.sh
cd soundeventdetection-pytorch_amd/csrc
rm -f *.o
make -j14 STAMPS=1 > /tmp/mk_st.log 2>&1 || (tail -20 /tmp/mk_st.log; exit 1)
cd ../..
for lb in 0 1; do
  echo "=== SED_BC_LB=$lb"
  SED_BC_LB=$lb timeout -k 10 200 python tools/bc_stamp.py 1 > /tmp/bc_stamp.$lb.log 2>&1 || { tail -5 /tmp/bc_stamp.$lb.log; }
  grep -E "bc producer|bc consumer" /tmp/bc_stamp.$lb.log | tail -4
done
