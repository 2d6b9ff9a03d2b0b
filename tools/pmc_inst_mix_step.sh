#!/bin/bash
# Dynamic instruction mix of every kernel of the bench step (rocprofv3 --pmc SQ_INSTS_*; counters only, no trace domains), summarised by
# tools/inst_mix.py.  The program itself follows `--`.   usage: tools/pmc_inst_mix_step.sh [tag]
set -e
tag=${1:-r05}
out=$PWD/gpurun_out/pmc_mix
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA -d $out/mix -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/${tag}_mix.err
python3 tools/inst_mix.py $(find $out/mix -name "${tag}*counter_collection.csv" | head -1) 1.9 > $out/${tag}_inst_mix.txt
head -40 $out/${tag}_inst_mix.txt
