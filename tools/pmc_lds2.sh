#!/bin/bash
# second LDS counter pass (GPU box): stall reasons beside the bank conflicts of tools/pmc_lds.sh.  usage: tools/pmc_lds2.sh [tag]
tag=${1:-r04}
out=$PWD/gpurun_out/prof_lds2
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_LDS_[A-Z_]*\|SQ_INSTS_LDS[A-Z_]*\|SQ_INST_LEVEL_LDS" | sort -u > $out/avail.txt
cat $out/avail.txt | tr '\n' ' '; echo
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" -d $out/$name -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/${name}.err || { tail -5 $out/${name}.err; return 1; }
}
run a SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES GRBM_GUI_ACTIVE && \
run b SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
for n in a b; do f=$(find $out/$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 tools/pmc_table.py $f SQ_BUSY_CYCLES; done
