#!/bin/bash
# M5's 3x3-embedded weight gradients (W = 8 interleaved frames): how much of a launch is the consumers' MFMA loop?  make DEBUG_SWITCHES=1
# (in-tree), SED_DBG=2 removes it (tools/ablate.sh); the product build is restored by the EXIT trap of tools/lib_restore.sh.
# (Upper bound for a middle-column-only weight gradient.)
set -e
source tools/lib_restore.sh
cd soundeventdetection-pytorch_amd/csrc
rm -f *.o
make -j14 DEBUG_SWITCHES=1 > /tmp/mk_dbg.log 2>&1 || (tail -20 /tmp/mk_dbg.log; exit 1)
cd ../..
for v in 0 2; do echo "== SED_DBG=$v"; SED_DBG=$v python tools/m5_breakdown.py 2>&1 | grep -E "wgrad_fused:|total"; done
