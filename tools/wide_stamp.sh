#!/bin/bash
# Phase stamps of the wide weight-gradient kernel (csrc/sed_wgrad_wide.hip) at the bench geometry.  The STAMPS=1 build is made in a
# SCRATCH COPY of the tree (/tmp/sed_stamp_tree) and run from there: the product libsed_hip.so is never touched.
# usage: tools/wide_stamp.sh [extra make flags, e.g. CXXFLAGS_EXTRA=-DSED_WIDE_X=1]
set -e
SRC=$(pwd)
DST=/tmp/sed_stamp_tree
rm -rf $DST && mkdir -p $DST
cp -r $SRC/include $SRC/tools $SRC/sed_amd.py $DST/
mkdir -p $DST/soundeventdetection-pytorch_amd
(cd $SRC/soundeventdetection-pytorch_amd && tar cf - --exclude='*.o' --exclude='*.so' --exclude='__pycache__' .) | (cd $DST/soundeventdetection-pytorch_amd && tar xf -)
cd $DST/soundeventdetection-pytorch_amd/csrc
make -j14 STAMPS=1 "$@" > /tmp/mk_wide_st.log 2>&1 || (tail -20 /tmp/mk_wide_st.log; exit 1)
cd $DST
timeout -k 10 300 python tools/ab_wgrad_wide.py 1 ${WIDE_STAMP_SET:-bench} 2>&1 | grep -E "^wide W|median"
