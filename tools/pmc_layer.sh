#!/bin/bash
# PMC counters of one layer's conv kernels (GPU box).  usage: tools/pmc_layer.sh KERNEL "B H W Cin Cout" -> gpurun_out/pmc_layer/
k=$1; shape=$2
out=$PWD/gpurun_out/pmc_layer
mkdir -p $out
export TMPDIR=/tmp
export SED_CONV_KERNEL=$k
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  -d $out/a_$k -o r --output-format csv -- python3 tools/bench_layer.py $shape 3 > $out/a_$k.log 2>&1 || { tail -5 $out/a_$k.log; exit 1; }
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES \
  -d $out/b_$k -o r --output-format csv -- python3 tools/bench_layer.py $shape 3 > $out/b_$k.log 2>&1 || { tail -5 $out/b_$k.log; exit 1; }
find $out -name "*counter_collection.csv"
