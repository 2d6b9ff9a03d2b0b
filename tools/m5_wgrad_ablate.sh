#!/bin/bash
# M5's 3x3-embedded weight gradients (W = 8 interleaved frames): how much of a launch is the consumers' MFMA loop?  make DEBUG_SWITCHES=1,
# SED_DBG=2 removes it (tools/ablate.sh); product build restored afterwards.  (Upper bound for a middle-column-only weight gradient.)
set -e
cd soundeventdetection-pytorch_amd/csrc
cp ../libsed_hip.so /tmp/libsed_hip.so.keep
rm -f *.o
make -j14 DEBUG_SWITCHES=1 > /tmp/mk_dbg.log 2>&1 || (tail -20 /tmp/mk_dbg.log; exit 1)
cd ../..
for v in 0 2; do echo "== SED_DBG=$v"; SED_DBG=$v python tools/m5_breakdown.py 2>&1 | grep -E "wgrad_fused:|total"; done
cd soundeventdetection-pytorch_amd/csrc && rm -f *.o && make -j14 > /tmp/mk_dbg2.log 2>&1 && cd ../..
cmp soundeventdetection-pytorch_amd/libsed_hip.so /tmp/libsed_hip.so.keep && echo "product build restored"
