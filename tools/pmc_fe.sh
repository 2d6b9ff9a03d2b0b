#!/bin/bash
# PMC view of the log-mel front-end kernels (tools/fe_time.py runs all three): instruction mix, then wait / busy counters.
set -e
out=$PWD/gpurun_out/pmc_fe
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA -d $out/mix -o fe --output-format csv -- python3 tools/fe_time.py > $out/mix.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $out/wait -o fe --output-format csv -- python3 tools/fe_time.py > $out/wait.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d $out/lds -o fe --output-format csv -- python3 tools/fe_time.py > $out/lds.log 2>&1 || true
python3 tools/inst_mix.py $(find $out/mix -name "*counter_collection.csv" | head -1) 2.1 > $out/fe_inst_mix.txt
python3 tools/pmc_table.py $(find $out/wait -name "*counter_collection.csv" | head -1) > $out/fe_wait.txt || true
python3 tools/pmc_table.py $(find $out/lds -name "*counter_collection.csv" | head -1) > $out/fe_lds.txt || true
cat $out/fe_inst_mix.txt $out/fe_wait.txt $out/fe_lds.txt | grep -i -E "frontend|kernel|name" | head -40
