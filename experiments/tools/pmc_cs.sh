#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x2 on gfx950) of the blocks 2-3 backward: two-kernel
# form vs the cin-sliced fused kernel (tools/ab_fused_cs.py).  The program itself follows `--`.
out=$PWD/gpurun_out/r04b_pmc_cs; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o cs --output-format csv -- python3 tools/ab_fused_cs.py 1 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o cs --output-format csv -- python3 tools/ab_fused_cs.py 1 > $out/write.log 2>&1
f=$(find $out/fetch -name "cs*counter_collection.csv" | head -1); w=$(find $out/write -name "cs*counter_collection.csv" | head -1)
python3 - "$f" "$w" <<'PY' | tee $out/../r04b_pmc_cs.txt
import sys, re
sys.path.insert(0, "tools")
from hbm_traffic import per_kernel
fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
rows = []
for k in fetch:
    if not re.search(r"conv_wgrad3_kernel<(16|8),|conv_pc_kernel<(16|8), 128, 0, (2|4)|conv_bwd_fused_cs_kernel", k):
        continue
    rd, wr = fetch[k] * 1024.0 * 2.0, write.get(k, 0.0) * 1024.0
    rows.append((k[:110], rd, wr))
print("HBM bytes per launch, averaged over the launches of a kernel name (B = 32): read (FETCH_SIZE x2), write, total [GB]")
for k, rd, wr in sorted(rows):
    print(f"{rd / 1e9:7.3f} {wr / 1e9:7.3f} {(rd + wr) / 1e9:7.3f}  {k}")
PY
