#!/bin/bash
# LDS bank-conflict counters of the other model families' kernels (CRNN recurrence, M5, default-width CNN): tools/bench_models.py under
# rocprofv3 --pmc (counters only).  usage: tools/pmc_models_lds.sh [tag]
tag=${1:-r04}
out=$PWD/gpurun_out/prof_models_lds
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/a -o $tag --output-format csv -- python3 tools/bench_models.py 2 > $out/a.log 2> $out/a.err || { tail -5 $out/a.err; exit 1; }
f=$(find $out/a -name "*counter_collection.csv" | head -1)
python3 tools/pmc_table.py $f SQ_BUSY_CYCLES
