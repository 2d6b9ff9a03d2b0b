#!/bin/bash
# Phase stamps of the split-operand kernels (csrc/sed_conv_x3.hip) at the bench geometry.  The STAMPS=1 build is made in a SCRATCH COPY of
# the tree (/tmp/sed_stamp_tree) and run from there: the product libsed_hip.so is never touched.
# usage: tools/x3_stamp.sh [layer substring] [extra make flags]
set -e
SRC=$(pwd)
DST=/tmp/sed_stamp_tree
pat=${1:-b1c2}; shift 1 2>/dev/null || true
rm -rf $DST && mkdir -p $DST
cp -r $SRC/include $SRC/tools $SRC/sed_amd.py $DST/
mkdir -p $DST/soundeventdetection-pytorch_amd
(cd $SRC/soundeventdetection-pytorch_amd && tar cf - --exclude='*.o' --exclude='*.so' --exclude='__pycache__' --exclude='ab' .) | (cd $DST/soundeventdetection-pytorch_amd && tar xf -)
cd $DST/soundeventdetection-pytorch_amd/csrc
make -j14 STAMPS=1 "$@" > /tmp/mk_x3_st.log 2>&1 || (tail -20 /tmp/mk_x3_st.log; exit 1)
cd $DST
SED_DBG=16 timeout -k 10 300 python tools/x3_layer_time.py "$pat" 1 2>&1 | grep -E "wgrad_x3|conv_x3|TF/s" | sort | uniq -c | sort -rn | head -40
