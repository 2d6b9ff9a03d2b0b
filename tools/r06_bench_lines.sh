#!/bin/bash
# Round 6 (GPU box): the bench-schema lines VERDICT round 5 asked for beside the headline -- measured peaks inside the roofline block,
# the reference's own shapes (REF-NATIVE front-end T = 182 at batch 32; batch 128 x 30-frame crops, main.py:110), eager and as a HIP-graph
# replay, the RCCL world-1 line with the communication fields and the all-reduce-only line.  Output: gpurun_out/r06_lines/ (copy to profiles/).
# usage: tools/r06_bench_lines.sh [tag]
set -e
tag=${1:-r06_a}
out=$PWD/gpurun_out/r06_lines
mkdir -p $out
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "headline done"
python3 bench.py --frontend ref_native --steps 200 --warmup 20 > $out/${tag}_bench_ref_native_T182.json 2> $out/${tag}_bench_ref_native_T182.err
python3 bench.py --frontend ref_native --steps 200 --warmup 20 --graph 1 --no-cpu-baseline > $out/${tag}_bench_ref_native_T182_graph.json 2> $out/${tag}_bench_ref_native_T182_graph.err
echo "ref-native done"
python3 bench.py --features-only --batch 128 --frames 30 --steps 200 --warmup 20 > $out/${tag}_bench_ref_crops_B128_T30.json 2> $out/${tag}_bench_ref_crops_B128_T30.err
python3 bench.py --features-only --batch 128 --frames 30 --steps 200 --warmup 20 --graph 1 --no-cpu-baseline > $out/${tag}_bench_ref_crops_B128_T30_graph.json 2> $out/${tag}_bench_ref_crops_B128_T30_graph.err
echo "ref crops done"
SED_DDP_FORCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --no-cpu-baseline > $out/${tag}_bench_rccl_world1.json 2> $out/${tag}_bench_rccl_world1.err
SED_DDP_FORCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --allreduce-only --steps 100 > $out/${tag}_allreduce_only_rccl_world1.json 2> $out/${tag}_allreduce_only_rccl_world1.err
echo "rccl world-1 done"
python3 tools/show_bench.py $out/${tag}_bench.json || true
ls -la $out
