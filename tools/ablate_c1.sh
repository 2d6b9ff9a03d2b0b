#!/bin/bash
# Ablation of block 0's C1-mode kernels inside the whole step (GPU box; numerically meaningless with SED_DBG != 0).
# conv_pc bits: 1 no output stores, 2 no MFMA loop, 4 no C1 build, 8 no global loads; wgrad bits: 1 no dz_out stores,
# 2 no consumer work, 8 no loads, 16 MFMAs without LDS reads, 32 LDS reads without MFMAs.
for dbg in 0 1 2 4 6 7 8 16 32; do
  SED_DBG=$dbg python bench.py --no-cpu-baseline --steps 5 --warmup 2 > /tmp/abl.json 2>/dev/null
  echo "DBG=$dbg $(python tools/show_bench.py /tmp/abl.json | grep -E 'fwd_c1|wgrad_fused_c1' | awk '{printf "%s %s | ", $1, $4}')"
done
