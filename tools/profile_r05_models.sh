#!/bin/bash
# Round-5 rocprofv3 evidence for BASELINE.json configs 4 and 5 (the CRNN at 16 clips per GPU, the raw-waveform M5 at 2880 frames): kernel
# statistics, HBM traffic by kernel (separate FETCH_SIZE / WRITE_SIZE passes, FETCH x2 on gfx950), MFMA utilisation, and a bench.py-schema
# JSON line per model.  The program itself follows `--` (python3 tools/bench_models.py ...): no env / bash -c hop under the profiler.
# Summaries land in gpurun_out/prof_r05_models/ (copy the ones to keep into profiles/).   usage: tools/profile_r05_models.sh [crnn|m5 ...]
set -e
out=$PWD/gpurun_out/prof_r05_models
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for m in ${@:-crnn m5}; do
  rocprofv3 --kernel-trace --stats -d $out/${m}_stats -o $m --output-format csv -- python3 tools/bench_models.py --json $m --steps 10 --warmup 3 --no-timer-pass > $out/r05_${m}_under_rocprof.json 2> $out/${m}_stats.err
  cp $(find $out/${m}_stats -name "${m}*kernel_stats.csv" | head -1) $out/r05_${m}_kernel_stats.csv
  echo "$m stats done"
  rocprofv3 --pmc FETCH_SIZE -d $out/${m}_fetch -o $m --output-format csv -- python3 tools/bench_models.py --json $m --steps 2 --warmup 2 --no-timer-pass > /dev/null 2> $out/${m}_fetch.err
  rocprofv3 --pmc WRITE_SIZE -d $out/${m}_write -o $m --output-format csv -- python3 tools/bench_models.py --json $m --steps 2 --warmup 2 --no-timer-pass > /dev/null 2> $out/${m}_write.err
  echo "$m fetch / write done"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d $out/${m}_mfma -o $m --output-format csv -- python3 tools/bench_models.py --json $m --steps 2 --warmup 2 --no-timer-pass > /dev/null 2> $out/${m}_mfma.err
  f=$(find $out/${m}_fetch -name "${m}*counter_collection.csv" | head -1); w=$(find $out/${m}_write -name "${m}*counter_collection.csv" | head -1)
  t=$(find $out/${m}_stats -name "${m}*kernel_trace.csv" | head -1); c=$(find $out/${m}_mfma -name "${m}*counter_collection.csv" | head -1)
  python3 tools/hbm_by_kernel.py $f $w $t $out/r05_${m}_hbm_by_kernel.json > $out/r05_${m}_hbm_traffic.txt
  python3 tools/mfma_util.py $c $t > $out/r05_${m}_mfma_util_pmc.txt
  cp $out/r05_${m}_hbm_by_kernel.json profiles/r05_${m}_hbm_by_kernel.json       # the JSON line below joins its labels with this table
  python3 tools/bench_models.py --json $m > $out/r05_${m}_bench.json 2> $out/${m}_bench.err
  echo "$m done"
done
ls -la $out | head -40
