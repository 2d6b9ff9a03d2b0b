#!/bin/bash
# Ablation sweep of block 0's forward (conv_pc_kernel<64,32,PRO_C1,EPI_STATS>, round-5 form without the mask) at the bench geometry:
# make DEBUG_SWITCHES=1 (in-tree; product build restored by the EXIT trap of tools/lib_restore.sh), SED_DBG bits 1 = no output stores,
# 2 = no MFMA k loop, 4 = no conv1 rebuild, 8 = no global loads (dead descriptors), 32 = no flush (staging reads, statistics, stores),
# 64 = no input-tile write / load issue, 128 = no staging writes.  Numerically meaningless for SED_DBG != 0.
set -e
source tools/lib_restore.sh
cd soundeventdetection-pytorch_amd/csrc
rm -f *.o
make -j14 DEBUG_SWITCHES=1 "$@" > /tmp/mk_dbg.log 2>&1 || (tail -20 /tmp/mk_dbg.log; exit 1)
cd ../..
for dbg in ${ABL_SET:-0 1 2 4 8 3 5 6 9 7 14 15 0}; do
  echo "SED_DBG=$dbg  $(PC_STAMP_NOMASK=1 SED_DBG=$dbg timeout -k 10 120 python tools/pc_stamp.py c1 2>&1 | grep 'ms per launch')"
done
