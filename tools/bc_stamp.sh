#!/bin/bash
# in-kernel phase stamps of the block-0 fused backward (builds STAMPS=1 into a scratch copy of csrc, leaves the product build alone)
set -e
rm -rf /tmp/csrc_st && cp -r soundeventdetection-pytorch_amd/csrc /tmp/csrc_st && cp -r include /tmp/ 2>/dev/null || true
mkdir -p /tmp/x/y && cp -r include /tmp/x/ 2>/dev/null || true
cd soundeventdetection-pytorch_amd/csrc
cp ../libsed_hip.so /tmp/libsed_hip.so.keep
rm -f *.o
make -j14 STAMPS=1 > /tmp/mk_st.log 2>&1 || (tail -20 /tmp/mk_st.log; exit 1)
cd ../..
for lb in 0 1; do
  echo "=== SED_BC_LB=$lb"
  SED_BC_LB=$lb timeout -k 10 200 python tools/bc_stamp.py 1 > /tmp/bc_stamp.$lb.log 2>&1 || { tail -5 /tmp/bc_stamp.$lb.log; }
  grep -E "bc producer|bc consumer" /tmp/bc_stamp.$lb.log | tail -4
done
cd soundeventdetection-pytorch_amd/csrc && rm -f *.o && make -j14 > /tmp/mk_st2.log 2>&1 && cd ../..
cmp soundeventdetection-pytorch_amd/libsed_hip.so /tmp/libsed_hip.so.keep && echo "product build restored"
