#!/bin/bash
# phase stamps of block 1's fused backward launches at the bench geometry (make STAMPS=1), product build restored afterwards
set -e
cd soundeventdetection-pytorch_amd/csrc
cp ../libsed_hip.so /tmp/libsed_hip.so.keep
rm -f *.o
make -j14 STAMPS=1 > /tmp/mk_st.log 2>&1 || (tail -20 /tmp/mk_st.log; exit 1)
cd ../..
timeout -k 10 300 python tools/ab_fused.py 1 3 2>&1 | grep -E "bf producer|bf consumer|fused" | tail -12
cd soundeventdetection-pytorch_amd/csrc && rm -f *.o && make -j14 > /tmp/mk_st2.log 2>&1 && cd ../..
cmp soundeventdetection-pytorch_amd/libsed_hip.so /tmp/libsed_hip.so.keep && echo "product build restored"
