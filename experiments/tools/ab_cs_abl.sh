#!/bin/bash
# ablations of csrc/sed_bwd_fused_cs.hip (wrong results, timing only): SED_CS_ABL bit 0 = no operator stream, bit 1 = dz arithmetic in slice 0 only.
# The kernel is an EXPERIMENTS=1 kernel; the product build is restored (and compared) by the EXIT trap of tools/lib_restore.sh.
mkdir -p gpurun_out/r04b
source tools/lib_restore.sh
cd soundeventdetection-pytorch_amd/csrc
for abl in 0 1 2 3; do
  rm -f *.o
  make -j14 EXPERIMENTS=1 CXXFLAGS_EXTRA="-DSED_CS_ABL=$abl" > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== SED_CS_ABL=$abl"
  (cd ../.. && timeout -k 10 200 python experiments/tools/ab_fused_cs.py 3 2>&1 | grep -E "fused|sum")
done
