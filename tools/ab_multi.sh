#!/bin/bash
# Several build variants against the default build on ONE box, each interleaved with the default: tools/ab_multi.sh "<cmd>" "<make args 1>" "<make args 2>" ...
set -e
source tools/lib_restore.sh      # EXIT trap: the product build comes back (and is compared) whatever happens below
cmd=$1; shift
cd soundeventdetection-pytorch_amd/csrc
run() { rm -f *.o; eval make -j14 $1 > /tmp/mk.log 2>&1 || (tail -20 /tmp/mk.log; exit 1); echo "== [$1]"; (cd ../.. && eval "$cmd" 2>&1 | grep -v amdgpu.ids); }
for args in "$@"; do run ""; run "$args"; done
run ""
