#!/bin/bash
# rocprofv3 evidence for bench.py's default workload (GPU box).  Writes under gpurun_out/prof/:
#   stats/   kernel trace + per-kernel statistics (rocprofv3 --kernel-trace --stats)
#   fetch/, write/  HBM traffic counters, one pass each (FETCH_SIZE / WRITE_SIZE cannot share a pass)
# usage: tools/profile.sh [tag]
set -e
tag=${1:-r01}
out=$PWD/gpurun_out/prof
mkdir -p $out
export TMPDIR=/tmp
args="bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $out/stats -o $tag --output-format csv -- python3 $args > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err
pargs="bench.py --steps 3 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o $tag --output-format csv -- python3 $pargs > /dev/null 2> $out/${tag}_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $out/write -o $tag --output-format csv -- python3 $pargs > /dev/null 2> $out/${tag}_write.err
find $out -name "*.csv" | head -20
