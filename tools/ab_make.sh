#!/bin/bash
# A/B of two sets of make arguments for the whole library on ONE box (full rebuilds; the product build is restored by the EXIT trap of tools/lib_restore.sh):
#   tools/ab_make.sh "<make args A>" "<make args B>" "<command>"       e.g.  tools/ab_make.sh "NOSLP_FILES=" "" "python bench.py ..."
set -e
source tools/lib_restore.sh      # EXIT trap: the product build comes back (and is compared) whatever happens below
a=$1; b=$2; shift; shift
cd soundeventdetection-pytorch_amd/csrc
for v in A B A B; do
  rm -f *.o
  if [ $v = A ]; then args=$a; else args=$b; fi
  eval make -j14 $args > /tmp/mk.log 2>&1 || (tail -20 /tmp/mk.log; exit 1)
  echo "== $v ($args)"
  (cd ../.. && eval "$@" 2>&1 | grep -v amdgpu.ids)
done
