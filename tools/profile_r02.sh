#!/bin/bash
# Round-2 rocprofv3 evidence for bench.py's default workload (GPU box): kernel stats, HBM traffic (separate FETCH / WRITE
# passes), MFMA utilisation.  Summaries land in gpurun_out/prof_r02/ (copy the ones to keep into profiles/).
# usage: tools/profile_r02.sh [tag]
set -e
tag=${1:-r02}
out=$PWD/gpurun_out/prof_r02
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d $out/stats -o $tag --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/${tag}_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d $out/write -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/${tag}_write.err
echo "write done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
  -d $out/mfma -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/${tag}_mfma.err
rocprofv3 --kernel-trace -d $out/mfma_trace -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/${tag}_mfma_trace.err
echo "mfma done"
f=$(find $out/fetch -name "*counter_collection.csv" | head -1); w=$(find $out/write -name "*counter_collection.csv" | head -1)
# bench.py runs the K steps twice (timed + instrumented pass): 3 + 2 + 3 = 8 steps per profiled run
python3 tools/hbm_traffic.py $f $w $out/${tag}_hbm_traffic_pmc.json 8 > $out/${tag}_hbm_traffic.txt
m=$(find $out/mfma -name "*counter_collection.csv" | head -1); t=$(find $out/mfma_trace -name "*kernel_trace.csv" | head -1)
python3 tools/mfma_util.py $m $t > $out/${tag}_mfma_util_pmc.txt
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
ls -la $out | head -30
