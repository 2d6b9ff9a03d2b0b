#!/bin/bash
# STAMPS + EXPERIMENTS build of the role-split forward kernel in a scratch tree; ablations: 2 = no matrix loop, 4 = idle loaders
set -e
SRC=$(pwd); DST=/tmp/sed_stamp_tree
rm -rf $DST && mkdir -p $DST
cp -r $SRC/include $SRC/tools $SRC/sed_amd.py $SRC/experiments $DST/
mkdir -p $DST/soundeventdetection-pytorch_amd
(cd $SRC/soundeventdetection-pytorch_amd && tar cf - --exclude='*.o' --exclude='*.so' --exclude='__pycache__' --exclude='ab' .) | (cd $DST/soundeventdetection-pytorch_amd && tar xf -)
cd $DST/soundeventdetection-pytorch_amd/csrc
make -j14 STAMPS=1 EXPERIMENTS=1 > /tmp/mk_x3_st.log 2>&1 || (tail -20 /tmp/mk_x3_st.log; exit 1)
cd $DST
for d in 0 2 4; do
  echo "== SED_DBG=$d (2: no matrix loop, 4: idle loaders)"
  SED_X3_CONV=p SED_DBG=$d timeout -k 10 300 python tools/x3_layer_time.py "c2" 10 2>&1 | grep "TF/s" | sed "s/of 16-bit MFMA//" | cut -d" " -f1-13
done
