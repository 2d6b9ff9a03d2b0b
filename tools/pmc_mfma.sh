#!/bin/bash
# MFMA utilisation / wait profile per kernel (GPU box): one PMC pass over a short bench run.
# usage: tools/pmc_mfma.sh [tag]   -> gpurun_out/prof/mfma/<tag>_counter_collection.csv
set -e
tag=${1:-r01}
out=$PWD/gpurun_out/prof
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
  -d $out/mfma -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/${tag}_mfma.err || { tail -5 $out/${tag}_mfma.err; exit 1; }
rocprofv3 --kernel-trace -d $out/mfma_trace -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/${tag}_mfma_trace.err
ls $out/mfma $out/mfma_trace
