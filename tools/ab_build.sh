#!/bin/bash
# A/B of a compile-time switch on ONE box: builds the library with and without `-D<flag>=0` and runs a command after each build.
# usage: tools/ab_build.sh SED_C1_WREG "python tools/pc_stamp.py c1"      (GPU box; the product build is restored by the EXIT trap of tools/lib_restore.sh)
set -e
source tools/lib_restore.sh      # EXIT trap: the product build comes back (and is compared) whatever happens below
flag=$1; shift
cd soundeventdetection-pytorch_amd/csrc
for v in ${AB_VALUES:-0 1 0 1}; do
  rm -f *.o
  make -j14 CXXFLAGS_EXTRA="-D${flag}=${v}" > /tmp/mk.log 2>&1 || (tail -20 /tmp/mk.log; exit 1)
  echo "== ${flag}=${v}"
  (cd ../.. && eval "$@" 2>&1 | grep -v amdgpu.ids)
done
