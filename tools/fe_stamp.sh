#!/bin/bash
# phase stamps of the batched log-mel front-end (make STAMPS=1, in-tree); the product build is restored by the EXIT trap of tools/lib_restore.sh
set -e
source tools/lib_restore.sh
cd soundeventdetection-pytorch_amd/csrc
rm -f *.o
make -j14 STAMPS=1 "$@" > /tmp/mk_st.log 2>&1 || (tail -20 /tmp/mk_st.log; exit 1)
cd ../..
SED_FE_KERNEL=0 timeout -k 10 200 python tools/fe_time.py 2>&1 | grep -E "^fe wave|ms" | sort | uniq -c | sort -rn | head -12
