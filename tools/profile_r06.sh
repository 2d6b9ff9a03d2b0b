#!/bin/bash
# Round-6 evidence for bench.py's default workload and the lines beside it (GPU box): rocprofv3 kernel statistics, HBM traffic (separate
# FETCH / WRITE passes, FETCH x2 on gfx950), MFMA utilisation, dynamic instruction mix; then plain bench lines on the same box: headline,
# fp32 / f16x3 / bf16x3 (the exact modes), class-default widths, RCCL world-1 + all-reduce-only, the reference's own shapes (REF-NATIVE
# T = 182, batch 128 x 30-frame crops; eager and HIP-graph), CRNN and M5.  Summaries land in gpurun_out/prof_r06/ (copy the ones to
# keep into profiles/).  The program itself follows `--` (python3 bench.py ...): no env / bash -c hop under the profiler.
# usage: tools/profile_r06.sh [tag]
set -e
tag=${1:-r06_z}
out=$PWD/gpurun_out/prof_r06
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d $out/stats -o $tag --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-measured-peaks > /dev/null 2> $out/${tag}_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d $out/write -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-measured-peaks > /dev/null 2> $out/${tag}_write.err
echo "write done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
  -d $out/mfma -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-measured-peaks > /dev/null 2> $out/${tag}_mfma.err
rocprofv3 --kernel-trace -d $out/mfma_trace -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-measured-peaks > /dev/null 2> $out/${tag}_mfma_trace.err
echo "mfma done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA -d $out/mix -o $tag --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-measured-peaks > /dev/null 2> $out/${tag}_mix.err
python3 tools/inst_mix.py $(find $out/mix -name "${tag}*counter_collection.csv" | head -1) 1.9 > $out/${tag}_inst_mix_pmc.txt
echo "inst mix done"
f=$(find $out/fetch -name "${tag}*counter_collection.csv" | head -1); w=$(find $out/write -name "${tag}*counter_collection.csv" | head -1)
# bench.py runs the K steps twice (timed + instrumented pass): 3 + 2 + 3 = 8 steps per profiled run
python3 tools/hbm_traffic.py $f $w $out/${tag}_hbm_traffic_pmc.json 8 > $out/${tag}_hbm_traffic.txt
m=$(find $out/mfma -name "${tag}*counter_collection.csv" | head -1); t=$(find $out/mfma_trace -name "${tag}*kernel_trace.csv" | head -1)
python3 tools/mfma_util.py $m $t > $out/${tag}_mfma_util_pmc.txt
cp $(find $out/stats -name "${tag}*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
# the counter run just taken becomes the traffic table bench.py reads (stamped with the kernel-source sha): the bench line below carries it
cp $out/${tag}_hbm_traffic_pmc.json profiles/hbm_traffic_by_label.json
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "headline done"
python3 bench.py --precision fp32 --steps 10 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_fp32.json 2> $out/${tag}_bench_fp32.err
python3 bench.py --precision f16x3 --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_f16x3.json 2> $out/${tag}_bench_f16x3.err
python3 bench.py --precision bf16x3 --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_bf16x3.json 2> $out/${tag}_bench_bf16x3.err
python3 bench.py --config default --batch 16 --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_default_widths.json 2> $out/${tag}_bench_default_widths.err
echo "precision / width lines done"
SED_DDP_FORCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --no-cpu-baseline > $out/${tag}_bench_rccl_world1.json 2> $out/${tag}_bench_rccl_world1.err
SED_DDP_FORCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --allreduce-only --steps 100 > $out/${tag}_allreduce_only_rccl_world1.json 2> $out/${tag}_allreduce_only_rccl_world1.err
echo "rccl world-1 done"
python3 bench.py --frontend ref_native --steps 200 --warmup 20 > $out/${tag}_bench_ref_native_T182.json 2> $out/${tag}_bench_ref_native_T182.err
python3 bench.py --frontend ref_native --steps 200 --warmup 20 --graph 1 --no-cpu-baseline > $out/${tag}_bench_ref_native_T182_graph.json 2> $out/${tag}_bench_ref_native_T182_graph.err
python3 bench.py --features-only --batch 128 --frames 30 --steps 200 --warmup 20 > $out/${tag}_bench_ref_crops_B128_T30.json 2> $out/${tag}_bench_ref_crops_B128_T30.err
python3 bench.py --features-only --batch 128 --frames 30 --steps 200 --warmup 20 --graph 1 --no-cpu-baseline > $out/${tag}_bench_ref_crops_B128_T30_graph.json 2> $out/${tag}_bench_ref_crops_B128_T30_graph.err
echo "reference shapes done"
python3 tools/bench_models.py --json crnn > $out/${tag}_crnn_bench.json 2> $out/${tag}_crnn_bench.err
python3 tools/bench_models.py --json m5 > $out/${tag}_m5_bench.json 2> $out/${tag}_m5_bench.err
echo "models done"
python3 tools/roofline_table.py $out/${tag}_bench.json $out/${tag}_inst_mix_pmc.txt > $out/${tag}_roofline_table.md
ls -la $out | head -60
