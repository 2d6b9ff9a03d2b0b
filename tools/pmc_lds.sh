#!/bin/bash
# LDS / issue-pipe utilisation per kernel (GPU box): PMC passes over a short bench run (counters only, no tracing).
# usage: tools/pmc_lds.sh [tag]   -> gpurun_out/prof_lds/<pass>/<tag>_counter_collection.csv
tag=${1:-r02}
out=$PWD/gpurun_out/prof_lds
mkdir -p $out
export TMPDIR=/tmp
run() {   # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" -d $out/$name -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/${name}.err || { tail -5 $out/${name}.err; return 1; }
}
run lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE && \
run act SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
# (a third pass over the TCP / TA / TD counters aborted inside rocprofv3 on this image and then sat idle until gpurun's
#  silence limit: left out)
find $out -name "*.csv" | head; tail -3 $out/*.err
