#!/bin/bash
# Timing ablations of the split-operand kernels (csrc/sed_conv_x3.hip) at the bench geometry, in a STAMPS=1 build made in a SCRATCH COPY of
# the tree: SED_DBG=2 drops the matrix loops, SED_DBG=4 the split / staging phase (wrong results, timing only).  If a kernel's time is the
# SUM of the two ablated times its phases do not overlap; if it equals the larger one they do.
# usage: tools/x3_ablate.sh [layer substring]
set -e
SRC=$(pwd)
DST=/tmp/sed_stamp_tree
pat=${1:-c2}
rm -rf $DST && mkdir -p $DST
cp -r $SRC/include $SRC/tools $SRC/sed_amd.py $DST/
mkdir -p $DST/soundeventdetection-pytorch_amd
(cd $SRC/soundeventdetection-pytorch_amd && tar cf - --exclude='*.o' --exclude='*.so' --exclude='__pycache__' --exclude='ab' .) | (cd $DST/soundeventdetection-pytorch_amd && tar xf -)
cd $DST/soundeventdetection-pytorch_amd/csrc
make -j14 STAMPS=1 > /tmp/mk_x3_st.log 2>&1 || (tail -20 /tmp/mk_x3_st.log; exit 1)
cd $DST
for d in ${X3_ABLATE_SET:-0 2 4 6}; do
  echo "== SED_DBG=$d  (2: no matrix loop, 4: no split/staging, 8: operator staged once per workgroup)"
  SED_DBG=$d timeout -k 10 300 python tools/x3_layer_time.py "$pat" 10 2>&1 | grep "TF/s" | sed "s/of 16-bit MFMA//"
done
