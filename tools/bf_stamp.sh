#!/bin/bash
# in-kernel phase stamps of the fused backward kernel (build: make -C soundeventdetection-pytorch_amd/csrc STAMPS=1 EXPERIMENTS=1)
for v in ${1:-3 0}; do
  echo "=== SED_BF_VAR=$v"
  SED_BF_VAR=$v timeout -k 10 200 python tools/ab_fused.py 1 $v > /tmp/bf_stamp.$v.log 2>&1 || exit 1
  grep -E "fused" /tmp/bf_stamp.$v.log
  for pat in "bf producer: 98" "bf consumer: 98" "bf producer: 19" "bf consumer: 19"; do grep -E "$pat" /tmp/bf_stamp.$v.log | tail -2; done
done
