#!/bin/bash
# PMC passes over the f16x3 step (GPU box): matrix-pipe utilisation, LDS conflicts, instruction mix per kernel.
# usage: tools/pmc_x3.sh [tag]     (summaries in gpurun_out/pmc_x3/)
set -e
tag=${1:-r06_x3}
out=$PWD/gpurun_out/pmc_x3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="python3 bench.py --precision f16x3 --steps 3 --warmup 2 --no-cpu-baseline --no-measured-peaks"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/mfma -o $tag --output-format csv -- $B > /dev/null 2> $out/mfma.err
rocprofv3 --kernel-trace -d $out/trace -o $tag --output-format csv -- $B > /dev/null 2> $out/trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA -d $out/mix -o $tag --output-format csv -- $B > /dev/null 2> $out/mix.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $out/lds -o $tag --output-format csv -- $B > /dev/null 2> $out/lds.err || echo "lds pass failed"
python3 tools/inst_mix.py $(find $out/mix -name "${tag}*counter_collection.csv" | head -1) 1.9 > $out/${tag}_inst_mix_pmc.txt
python3 tools/mfma_util.py $(find $out/mfma -name "${tag}*counter_collection.csv" | head -1) $(find $out/trace -name "${tag}*kernel_trace.csv" | head -1) > $out/${tag}_mfma_util_pmc.txt
cp $(find $out/lds -name "${tag}*counter_collection.csv" | head -1) $out/${tag}_lds_counters.csv || true
echo done
