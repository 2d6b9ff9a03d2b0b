#!/usr/bin/env python3
"""Headline benchmark: SED *train* clips/sec of the MI355X pipeline on BASELINE.json configs[1]
(Cnn_AvgPooling main widths 32/64/128/128, bf16, synthetic 60 s / 32 kHz / 64-mel clips, batch 32
per GPU).  One step = log-mel front-end from the HBM-resident waveform batch -> forward -> weighted
BCE -> backward -> (RCCL gradient all-reduce) -> fused Adam-amsgrad.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` for the dominant
kernel (HIP-event timed inside the timed region, on the stream the kernels run on) and, at N=1,
`cpu_baseline` (the CPU oracle's ATen-autograd restatement of the same step, timed on the host)."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAIN_CFG = [(32, 2), (64, 2), (128, 2), (128, 1)]
DEFAULT_CFG = [(64, 2), (128, 2), (256, 2), (512, 1)]
PEAK_MFMA_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0 / 3, "f16x3": 2500.0 / 3}    # MI355X_MICROARCH.md, dense (x3: three 16-bit MFMAs per algorithmic product)
PEAK_HBM_GBS = 8000.0


def synth_wave(B, samples, sr, seed, device):
    """N(0, 0.1^2) clipped to [-1, 1] plus three Hann-enveloped 0.5 s bursts per clip."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    w = (torch.randn(B, samples, generator=g) * 0.1).clamp_(-1, 1)
    blen = sr // 2
    env = torch.hann_window(blen, periodic=False)
    t = torch.arange(blen) / sr
    for b in range(B):
        for _ in range(3):
            s = int(torch.randint(0, samples - blen, (1,), generator=g))
            f = float(torch.randint(300, 6000, (1,), generator=g))
            w[b, s:s + blen] += 0.5 * env * torch.sin(2 * np.pi * f * t)
    return w.clamp_(-1, 1).to(device)


def synth_targets(B, T, K, seed, device, p=0.04, run=10):
    g = np.random.default_rng(seed)
    y = np.zeros((B, T, K), dtype=np.float32)
    n_runs = max(1, int(round(p * T / (1.5 * run))))
    for b in range(B):
        for k in range(K):
            for _ in range(n_runs):
                s = int(g.integers(0, max(1, T - 2 * run)))
                y[b, s:s + run + int(g.integers(0, run)), k] = 1.0
    return torch.from_numpy(y).to(device)


def layer_costs(plan, engine, elem_bytes):
    """label -> (flops, SURVEY 8(d) bytes, kernel-dataflow bytes) for the conv launches of one step.

    8(d) bytes = the convolution's input and output tensor touched once per pass (forward: x + z; data gradient: dz + dx;
    weight gradient: x + dz): `roofline.frac_8d_convention` is computed from it.  Dataflow bytes = what the fused kernel as
    built has to move (extra operands of the fused BatchNorm / ReLU / pool backward, the dz it materialises, C1-mode
    substitutions): with the FLOPs they decide `roofline.bound` / `frac` (replaced by the PMC traffic when the committed
    counter run was taken with the tree's kernel sources)."""
    costs = {}
    B = plan.B
    for bi, blk in enumerate(plan.layers):
        for j, ly in enumerate(blk):
            px = B * ly.H * ly.W
            flops = 2.0 * 9 * ly.cin * ly.cout * px
            in_b = px * (4 if ly.cinp == 1 else ly.cinp * elem_bytes)
            out_b = px * ly.coutp * elem_bytes
            alg = in_b + out_b
            tag = f"b{bi}c{j + 1} {ly.cin}->{ly.cout} H{ly.H} W{ly.W}"
            first = ly.cinp == 1
            costs[("sed_conv3x3_c1_fwd" if first else "sed_conv3x3_fwd") + ":fwd " + tag] = (flops, alg, alg)
            costs[("sed_conv3x3_c1_wgrad" if first else "sed_conv3x3_wgrad") + ":bwd " + tag] = (flops, alg, alg)
            if not first:   # fused form: reads x, z and g (c2: pooled g = 1/4), writes dz
                pool = engine.cfg[bi][1]
                g_b = out_b / (pool * pool) if j == 1 else out_b
                costs["sed_conv3x3_wgrad_fused:bwd " + tag] = (flops, alg, in_b + out_b + g_b + out_b)
            if not first:   # data gradient (the c2 one also re-reads z1 for the fused ReLU/BN epilogue)
                extra = in_b if j == 1 else 0
                costs["sed_conv3x3_fwd:bwd " + tag] = (flops, alg, in_b + out_b + extra)
                if j == 0:  # conv1's data gradient with the pooled-tensor statistics of the previous block in its epilogue:
                    # + pooled activation (bf16) and active-pixel counts (1 B) at the resolution of its output dy
                    costs["sed_conv3x3_dgrad_poolstats:bwd " + tag] = (flops, alg, in_b + out_b + in_b + in_b / elem_bytes)
            if not first:
                # Fused weight + data gradient (csrc/sed_bwd_fused.hip, round 3): the launch performs BOTH backward passes of the
                # layer.  8(d) bytes: each tensor the two passes touch counted ONCE (x, dz, dx -- the shared dz is not counted
                # twice although the kernel never moves it at all); dataflow = what it really reads / writes.
                pool = engine.cfg[bi][1]
                g_b = out_b / (pool * pool) if j == 1 else out_b
                ref_b = 0 if j == 1 else in_b / elem_bytes           # c1: active-pixel counts (1 B/element); c2: z1 is read once
                costs["sed_conv3x3_bwd_fused:bwd " + tag] = (2 * flops, in_b + out_b + in_b, in_b + out_b + g_b + ref_b + in_b)
            if bi == 0 and j == 1:
                # block 0, C1 mode: the gated data gradient is contracted in registers (never written): reads the fp32 input, z2,
                # the pooled dy and the mask; writes nothing per pixel
                pool0 = engine.cfg[bi][1]
                costs["sed_conv3x3_bwd_fused_c1:bwd " + tag] = (2 * flops + 2.0 * 10 * 32 * px, in_b + out_b + in_b,
                                                                px * 4 + out_b + out_b / (pool0 * pool0) + px * 4)
            if bi == 0 and j == 1:
                # "C1 mode" (block 0 without conv1's output in memory): the 1-channel fp32 input (4 B/pixel) replaces z1,
                # a 4 B/pixel bit mask of conv1's ReLU decisions is written by the forward and read by the data gradient
                pool = engine.cfg[bi][1]
                costs["sed_conv3x3_fwd_c1:fwd " + tag] = (flops, alg, px * 4 + out_b + px * 4)
                costs["sed_conv3x3_wgrad_fused_c1:bwd " + tag] = (flops, alg, px * 4 + out_b + out_b / (pool * pool) + out_b)
                costs["sed_conv3x3_dgrad_c1:bwd " + tag] = (flops, alg, out_b + px * 4 + in_b)
                # fused form (csrc/sed_dgrad_c1.hip): reads dz2, the 1-channel input and the mask; g is never written
                costs["sed_conv3x3_dgrad_c1_stats:bwd " + tag] = (flops + 2.0 * 10 * 32 * px, alg, out_b + px * 4 + px * 4)
    return costs


def roof_of(flops, byts, seconds, precision):
    """One launch against min(P_mfma, AI * BW_hbm): dict(ms, tflops, gbs, ai, bound, roof_tflops, frac)."""
    peak_f, peak_b = PEAK_MFMA_TFLOPS[precision] * 1e12, PEAK_HBM_GBS * 1e9
    ai = flops / byts if byts else 0.0
    roof = min(peak_f, ai * peak_b) if flops else 0.0
    d = {"ms": seconds * 1e3, "tflops": flops / seconds / 1e12, "gbs": byts / seconds / 1e9, "ai": ai,
         "bound": "mfma" if ai * peak_b >= peak_f else "hbm"}
    if flops:
        d.update(roof_tflops=roof / 1e12, frac=flops / seconds / roof)
    else:
        d.update(roof_tflops=None, frac=byts / seconds / peak_b)
    return d


def physical_roof(flops, phys_bytes, seconds, precision):
    """The launch AS BUILT against the roof that binds it: max(FLOPs / P_mfma, bytes it really moves / BW_hbm) over its
    measured time.  `phys_bytes` = PMC traffic when a fresh counter run exists, else the kernel's dataflow bytes (what it
    must read and write as fused).  Returns (bound, achieved, peak, unit, frac)."""
    peak_f, peak_b = PEAK_MFMA_TFLOPS[precision] * 1e12, PEAK_HBM_GBS * 1e9
    t_f, t_b = flops / peak_f, phys_bytes / peak_b
    if flops and t_f >= t_b:
        return "mfma", flops / seconds / 1e12, PEAK_MFMA_TFLOPS[precision], "TFLOP/s", t_f / seconds
    return "hbm", phys_bytes / seconds / 1e9, PEAK_HBM_GBS, "GB/s", t_b / seconds


def load_traffic(B, T, precision, config):
    """profiles/hbm_traffic_by_label.json (committed PMC run of the default workload) -> (table, sha, stale).  The table is only
    valid for the workload it was measured on AND for the kernel sources it was measured with (tools/csrc_sha.py)."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic_by_label.json")) as f:
            tab = json.load(f)
    except (OSError, ValueError):
        return None, None, None
    if not (B == 32 and T == 6001 and precision == "bf16" and config == "main"):
        return None, tab.get("__csrc_sha256__"), None
    from tools.csrc_sha import csrc_sha256
    sha = tab.get("__csrc_sha256__")
    stale = sha != csrc_sha256(ROOT)
    return (None if stale else tab), sha, stale


def measure_peaks(sed, dev, mfma_iters=4000, copy_bytes=1 << 30):
    """Box-measured peaks, taken once before the timed region with the library's own micro-kernels
    (csrc/sed_peaks.hip): a register-fed v_mfma_f32_32x32x16_bf16 loop on pseudo-random operands and a float4 stream copy over
    `copy_bytes` (read + write).  HIP events on the launch stream; best of three launches each, after one warm-up launch."""
    import ctypes
    L = sed._lib
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    sink = torch.zeros(1, device=dev)
    fl = ctypes.c_double(0.0)

    def timed(fn, reps=3):
        best = None
        for i in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            if i > 0 and (best is None or ms < best):
                best = ms
        return best

    ms_m = timed(lambda: L.check(lib.sed_peak_mfma_bf16(mfma_iters, L.ptr(sink), ctypes.addressof(fl), st), "peak_mfma"))
    src = torch.empty(copy_bytes // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    ms_c = timed(lambda: L.check(lib.sed_peak_stream_copy(L.ptr(src), L.ptr(dst), copy_bytes, st), "peak_copy"))
    ms_r = timed(lambda: L.check(lib.sed_peak_stream_read(L.ptr(src), copy_bytes, L.ptr(sink), st), "peak_read"))
    del src, dst
    return {"mfma_bf16_tflops": fl.value / (ms_m * 1e-3) / 1e12, "mfma_launch_ms": ms_m,
            "hbm_copy_gbs": 2.0 * copy_bytes / (ms_c * 1e-3) / 1e9, "copy_launch_ms": ms_c, "copy_bytes_each_way": copy_bytes,
            "hbm_read_gbs": copy_bytes / (ms_r * 1e-3) / 1e9, "read_launch_ms": ms_r,
            "hbm_peak_measured_gbs": max(2.0 * copy_bytes / (ms_c * 1e-3), copy_bytes / (ms_r * 1e-3)) / 1e9,
            "how": "csrc/sed_peaks.hip: register-fed v_mfma_f32_32x32x16_bf16 loop (4 waves/SIMD, pseudo-random operands) and a float4 "
                   "grid-stride copy (bytes read + bytes written) / read-only stream over 1 GiB; HIP events, best of 3 after a warm-up, before the timed region"}


def allreduce_times(dist, flat_g, buckets, iters, dev):
    """Each gradient bucket's all-reduce in ISOLATION (and the whole flat buffer as one): `iters` synchronous collectives back to back,
    HIP events on the current stream (torch.distributed orders a synchronous collective after the kernels already enqueued on it and
    the stream's later work after the collective).  Returns [(keys, floats, us per all-reduce)], us for the single flat one."""
    out = []
    spans = [(list(k), s, e) for k, s, e in buckets] + [(["<whole flat buffer>"], 0, flat_g.numel())]
    scratch = torch.zeros_like(flat_g)
    for keys, s, e in spans:
        buf = scratch[s:e]
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            dist.all_reduce(buf)
        e1.record()
        e1.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) * 1e3 / iters], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out.append({"groups": keys, "floats": int(e - s), "bytes": int(e - s) * 4, "us_per_allreduce_isolated": float(t.item())})
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) with torch.distributed.run as a CHILD
    process, relay rank 0's single JSON line on this process's stdout, return the children's exit code.  The parent never
    initialises the GPU and never execs (either would be fatal on the GPU pool)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr=127.0.0.1",
           f"--master-port={port}", os.path.abspath(__file__), *argv]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = []
    for ln in p.stdout:
        if ln.startswith('{"metric"'):
            lines.append(ln)
        else:                      # anything else a rank (or librccl) wrote on stdout: keep it visible, off the result channel
            sys.stderr.write(ln)
    rc = p.wait()
    if rc == 0 and len(lines) != 1:
        sys.stderr.write(f"bench.py launcher: expected one result line from rank 0, got {len(lines)}\n")
        rc = 1
    if lines:
        sys.stdout.write(lines[-1])
        sys.stdout.flush()
    return rc


def rendezvous_rehearsal(a, world, rank):
    """--rendezvous-only: the multi-rank control flow of a bench run -- stdout handed to stderr while the group is alive, barrier, timed
    region, barrier, MAX over ranks, ONE line from rank 0 -- around an EMPTY step, over gloo, without touching the GPU."""
    import torch.distributed as dist
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    print(f"rank {rank}/{world}: rendezvous ok", flush=True)          # goes to stderr (fd 1 is redirected): must not reach the result channel
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(0.001 * (1 + rank))                                 # ranks finish at different times: the MAX must pick the slowest
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    mine = float(el[0])
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ok = torch.tensor([1 if float(el[0]) >= mine else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        line = {"metric": "SED train clips/sec (60s,64-mel,9-layer CNN)", "value": 0.0, "unit": "clips/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": float(el[0]) / max(1, a.steps) * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": a.precision, "data": "none",
                "rehearsal": "rendezvous only: no train step ran, no GPU was touched; value is not a measurement",
                "config": {"workload": "launcher rehearsal", "global_batch": a.batch * world, "parallelism": f"dp{world}",
                           "max_over_ranks_ok": bool(int(ok[0]))}}
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--config", default="main", choices=["main", "default"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "f16x3", "bf16x3"],
                    help="f16x3 / bf16x3: fp32 tensors, operands split into two fp16 / bf16 pieces on the 16-bit matrix pipe (csrc/sed_conv_x3.hip); f16x3 is the fast reference-exact mode")
    ap.add_argument("--no-frontend", "--features-only", dest="no_frontend", action="store_true",
                    help="time the CNN step on precomputed features")
    ap.add_argument("--frontend", default="bench", choices=["bench", "ref_native"],
                    help="front-end constants: bench = 32 kHz / NFFT 1024 / hop 320 (T = 6001 per 60 s, BASELINE config 2); ref_native = the "
                         "reference's committed 48 kHz / NFFT 32768 / hop 15840 (T = 182 per 60 s; dataset/spectogram/spectogram_configs.py:5-14)")
    ap.add_argument("--frames", type=int, default=0, help="with --features-only: frames per crop instead of a whole --seconds clip "
                    "(the reference trains on batch 128 x 30-frame crops: main.py:110, spectogram_configs.py:10)")
    ap.add_argument("--graph", type=int, default=0, help="1: FusedTrainer(graph=True): the step captured into a HIP graph and replayed "
                    "(single process; the reference-shape steps are ~90 dependent launches of a few microseconds each)")
    ap.add_argument("--allreduce-only", action="store_true", help="no train step: time the gradient all-reduces of the bucket layout (and "
                    "one flat all-reduce of the whole 2.33 MB buffer) in isolation, --steps iterations each, and print one JSON line")
    ap.add_argument("--no-measured-peaks", action="store_true", help="skip the two ~50 ms micro-kernels behind roofline.peak_measured")
    ap.add_argument("--overlap-frontend", type=int, default=0, help="1: front-end of the next batch on a second stream; 2: and the train "
                    "step on a high-priority stream; 3: the front-end on a LOW-priority stream (fills the CUs the small launches leave free)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--sync-bn", type=int, default=0, help="1: BatchNorm statistics over the global batch (all-reduce of the "
                                                           "per-layer sums); default: per-rank statistics")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets two ranks "
                                                      "share one GPU to test the multi-rank path)")
    ap.add_argument("--rendezvous-only", action="store_true", help="launcher rehearsal WITHOUT the GPU: the N ranks rendezvous over gloo, run "
                    "the barriers / MAX-reduce of the timed region around an empty step and rank 0 prints one line marked \"rehearsal\" "
                    "(tests/test_ddp_gloo.py runs the driver's 8-rank command shape this way; a GPU box admits 6 processes on its card, the test runner included)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it has not touched the GPU and never will)
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    import torch.distributed as dist
    if a.rendezvous_only:
        raise SystemExit(rendezvous_rehearsal(a, world, rank))
    dev_index = (local_rank % max(1, torch.cuda.device_count())) if world > 1 else 0
    # SED_DDP_FORCE=1 under torch.distributed.run --nproc-per-node 1: a world-size-1 RCCL group whose (identity) gradient
    # all-reduces really execute inside the timed step -- shows the collective kernels beside the persistent 256-workgroup
    # conv kernels on the one GPU a builder box has (profiles/r03_*_rccl_world1*)
    grouped = world > 1 or (os.environ.get("SED_DDP_FORCE", "0") == "1" and "RANK" in os.environ)
    json_fd = None
    if grouped:
        # librccl prints a version banner on the C-level stdout of rank 0 when the communicator comes up: the driver expects ONE JSON
        # line there.  Everything written to fd 1 while the group is alive goes to stderr; the result line is written to the saved fd.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(a.backend)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    sed = importlib.import_module("soundeventdetection-pytorch_amd")
    pp = importlib.import_module("soundeventdetection-pytorch_amd.dataset.spectogram.preprocess")
    sc = importlib.import_module("soundeventdetection-pytorch_amd.dataset.spectogram.spectogram_configs")
    cfg = MAIN_CFG if a.config == "main" else DEFAULT_CFG
    fcfg = sc.BENCH if a.frontend == "bench" else sc.REF_NATIVE
    samples = int(a.seconds * fcfg.working_sample_rate)
    T = fcfg.num_frames(samples)
    B = a.batch
    if a.frames:
        if not a.no_frontend:
            raise SystemExit("--frames needs --features-only (a crop of a log-mel table has no waveform of its own)")
        T = int(a.frames)

    torch.manual_seed(0)
    model = sed.Cnn_AvgPooling(1, cfg, precision=a.precision).to(dev)
    trainer = sed.FusedTrainer(model, lr=1e-6, recall_factor=5.0, sync_bn=bool(a.sync_bn), graph=bool(a.graph))     # main.py:107,111 defaults
    if a.allreduce_only:
        if not grouped:
            raise SystemExit("--allreduce-only needs a process group (torch.distributed.run, or --gpus N; one GPU: SED_DDP_FORCE=1 ... --nproc-per-node 1)")
        res = allreduce_times(dist, trainer.flat.g, trainer.flat.buckets, max(1, a.steps), dev)
        if rank == 0:
            line = {"metric": "gradient all-reduce latency (two-bucket layout of the 2.33 MB flat fp32 gradient buffer)", "unit": "us",
                    "value": res[-1]["us_per_allreduce_isolated"], "n_gpus": world, "steps": a.steps, "warmup": 5, "higher_is_better": False,
                    "backend": a.backend, "rccl_group": a.backend == "nccl", "buckets": res[:-1], "flat": res[-1],
                    "config": {"workload": "all-reduce only: no train step ran", "parallelism": f"dp{world}"}}
            os.write(json_fd, (json.dumps(line) + "\n").encode())
        dist.barrier()
        dist.destroy_process_group()
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        os.close(json_fd)
        return
    peaks = None
    if rank == 0 and not a.no_measured_peaks:
        peaks = measure_peaks(sed, dev)
    y = synth_targets(B, T, 1, 4321 + rank, dev)
    wave = None
    if a.frames:
        # feature-level synthetic input: standard-normal log-mel crops (the features are z-scored, a4)
        gfe = torch.Generator(device="cpu").manual_seed(1234 + rank)
        feats = torch.randn((B, 1, T, fcfg.mel_bins), generator=gfe).to(dev)
        mean = std = None
        fe = None
    else:
        wave = synth_wave(B, samples, fcfg.working_sample_rate, 1234 + rank, dev)
        # dataset statistics for the z-score (a4): from this rank's synthetic batch, computed once
        fe0 = pp.LogMelFrontEnd(fcfg, dev)
        raw = fe0(wave)
        mean = raw.mean(dim=(0, 1, 2))
        std = raw.std(dim=(0, 1, 2), unbiased=False)
        fe = pp.LogMelFrontEnd(fcfg, dev, mean=mean, std=std)
        feats = torch.empty((B, 1, T, fcfg.mel_bins), dtype=torch.float32, device=dev)
        fe(wave, out=feats)
        del raw
    comm_iso = None
    if grouped and trainer.reducer.enabled:
        comm_iso = allreduce_times(dist, trainer.flat.g, trainer.flat.buckets, 20, dev)

    # --overlap-frontend: the log-mel front-end of step i+1 runs on a second HIP stream beside the train step of batch i
    # (double-buffered features; every timed step still contains exactly one front-end pass and one train step)
    pf = None
    compute_stream = None
    if a.overlap_frontend and not a.no_frontend:
        side = None
        if a.overlap_frontend >= 2:
            # 2: the train step on a HIGH-priority stream, the front-end on a normal one; 3: the front-end on a LOW-priority
            # stream (hipStreamCreateWithPriority(+1) -- torch only hands out normal / high), the train step on the default one
            lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
            if a.overlap_frontend == 2:
                compute_stream = torch.cuda.Stream(device=dev, priority=-1)
            else:
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so")
                raw = ctypes.c_void_p()
                lo_p, hi_p = ctypes.c_int(), ctypes.c_int()
                hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo_p), ctypes.byref(hi_p))
                rc = hip.hipStreamCreateWithPriority(ctypes.byref(raw), 1, lo_p.value)      # 1 = hipStreamNonBlocking
                if rc != 0:
                    raise SystemExit(f"hipStreamCreateWithPriority failed: {rc}")
                print(f"front-end stream priority {lo_p.value} (device range least {lo_p.value} .. greatest {hi_p.value})", file=sys.stderr)
                side = torch.cuda.ExternalStream(raw.value, device=dev)
        if compute_stream is not None:
            torch.cuda.synchronize()
            torch.cuda.set_stream(compute_stream)
        pf = pp.PrefetchingFrontEnd(fe, stream=side)
        pf.submit(wave)

    def step():
        if pf is not None:
            x = pf.get()
            pf.timer = trainer.engine.timer
            pf.submit(wave)
            out = trainer.train_step(x, y)
            pf.release()
            return out
        if not a.no_frontend:
            if trainer.engine.timer is not None:   # same event pair + host-stall filter as the engine's launches
                trainer.engine.timer.launch("sed_logmel_fwd", lambda: fe(wave, out=feats), ())
            else:
                fe(wave, out=feats)
        return trainer.train_step(feats, y)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # per-kernel HIP-event timings: an instrumented pass of the same K steps right after the timed
    # region (two event records per launch would otherwise make the timed region host-bound)
    # (every rank runs it -- the steps contain the gradient collectives -- but only rank 0 records events)
    timer = None
    if rank == 0:
        timer = sed.engine.KernelTimer()
        trainer.engine.timer = timer
    inst_steps = a.steps
    if a.graph:
        # a replayed graph has no per-launch host calls to bracket: the instrumented pass runs the same step eagerly
        trainer.use_graph = False
        inst_steps = min(a.steps, 20)
    if trainer.reducer.enabled:
        trainer.reducer.exposed_events = []
    bn_calls0 = trainer.engine.bn_sync.calls if getattr(trainer.engine, "bn_sync", None) is not None else 0
    for _ in range(inst_steps):
        step()
    torch.cuda.synchronize()
    trainer.engine.timer = None
    exposed_ms = None
    if trainer.reducer.enabled:
        ev = trainer.reducer.exposed_events
        trainer.reducer.exposed_events = None
        exposed_ms = sum(e0.elapsed_time(e1) for e0, e1 in ev) / max(1, inst_steps)
    bn_calls = ((trainer.engine.bn_sync.calls - bn_calls0) / max(1, inst_steps)) if getattr(trainer.engine, "bn_sync", None) is not None else 0
    if grouped:
        dist.barrier()
    if grouped:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    loss_val = float(loss.item())

    result = None
    if rank == 0:
        clips = world * B * a.steps
        value = clips / elapsed
        # ---- per-kernel roofline of the dominant kernel -------------------------------------
        summ = timer.summary()
        n_all = {k: len(v) for k, v in timer.samples().items()}
        plan = next(iter(trainer.engine._plans.values()))
        eb = 2 if a.precision == "bf16" else 4
        costs = layer_costs(plan, trainer.engine, eb)
        fe_bytes = B * ((0 if a.frames else samples * 4) + T * fcfg.mel_bins * 4)
        costs["sed_logmel_fwd"] = (0.0, fe_bytes, fe_bytes)
        per_step = {k: t / n * n_all.get(k, n) / inst_steps for k, (n, t) in summ.items()}   # ms per step by label
        top = sorted(summ.items(), key=lambda kv: -per_step[kv[0]])
        dom_label, (dom_n, dom_ms) = top[0]
        roof = {"kernel": dom_label, "launches": dom_n, "avg_ms": dom_ms / dom_n,
                "share_of_gpu_time": per_step[dom_label] / sum(per_step.values()),
                "launches_dropped_as_host_stalls": getattr(timer, "dropped", {}),
                "timing": "HIP events around each launch on the launch stream, instrumented pass of the same "
                          "steps directly after the timed region"}
        traffic_tab, traffic_sha, traffic_stale = load_traffic(B, T, a.precision, a.config)
        if dom_label in costs:
            flops, byts, dflow = costs[dom_label]
            avg_s = dom_ms / dom_n / 1e3
            r = roof_of(flops, byts, avg_s, a.precision)
            ent = traffic_tab.get(dom_label) if traffic_tab else None
            traffic = ent["hbm_bytes_per_launch"] if ent else None
            # headline = the PHYSICAL fraction (round-3 verdict): FLOPs against the MFMA peak or the bytes the kernel really moves
            # against the HBM peak, whichever binds.  The SURVEY 8(d)-convention figure (tensor passes the two-kernel form would
            # touch; a fused launch moves fewer) is kept beside it as frac_8d_convention and never decides `bound`.
            bound, ach, peak, unit, frac = physical_roof(flops, traffic if traffic else dflow, avg_s, a.precision)
            roof.update({"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": frac,
                         "frac_definition": "max(algorithmic FLOPs / MFMA peak, HBM bytes moved / HBM peak) / measured launch time; "
                                            "bytes = PMC traffic when the committed counter run matches the tree, else kernel_dataflow_bytes",
                         "traffic": traffic, "traffic_stale": traffic_stale, "traffic_csrc_sha": traffic_sha,
                         "traffic_source": "profiles/hbm_traffic_by_label.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                                           "passes, gfx950 FETCH x2 correction; committed PMC run of this workload with these kernel "
                                           "sources -- sha checked --, not re-measured in this run)" if traffic else None,
                         "algorithmic_flops": flops, "algorithmic_bytes": byts,
                         "algorithmic_bytes_convention": "SURVEY 8(d): conv input + output tensor once per pass; a fused two-pass launch "
                                                         "(weight + data gradient) counts each tensor of its passes once (x, dz, dx)",
                         "frac_8d_convention": r["frac"], "bound_8d_convention": r["bound"],
                         "achieved_8d_convention_gbs": r["gbs"], "achieved_tflops": r["tflops"],
                         "kernel_dataflow_bytes": dflow, "kernel_dataflow_gbs": dflow / avg_s / 1e9,
                         "arithmetic_intensity_8d": r["ai"],
                         "arithmetic_intensity_physical": flops / (traffic if traffic else dflow)})
        else:
            roof.update({"bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None,
                         "traffic": None, "traffic_stale": traffic_stale, "traffic_csrc_sha": traffic_sha})
        # box-measured peaks beside the spec ones: the same achieved figure over what THIS part delivers to a register-fed
        # MFMA loop / a float4 stream copy (csrc/sed_peaks.hip, measured just before the timed region)
        pk_f = pk_b = None
        if peaks is not None:
            pk_f = peaks["mfma_bf16_tflops"] * (PEAK_MFMA_TFLOPS[a.precision] / PEAK_MFMA_TFLOPS["bf16"])
            pk_b = peaks["hbm_peak_measured_gbs"]          # the larger of the read-only and the copy stream: the conservative denominator
            roof["peaks_measured"] = peaks
            roof["peak_spec"] = roof.get("peak")
            if roof.get("achieved") is not None:
                roof["peak_measured"] = pk_f if roof["bound"] == "mfma" else pk_b
                roof["frac_of_measured"] = roof["achieved"] / roof["peak_measured"]
            if a.precision != "bf16":
                roof["peak_measured_note"] = "MFMA peak measured with the bf16 loop, scaled by the spec ratio of this dtype's dense peak"
        # every conv launch against the roof that binds it as built (bytes: PMC traffic if fresh, else dataflow), with the
        # 8(d)-convention figure min(P_mfma, AI_8d * BW_hbm) beside it (north_star)
        layer_roof = {}
        for k, (n, t) in top:
            if k in costs and costs[k][0] > 0:
                fl, by, df = costs[k]
                sec = t / n / 1e3
                d8 = roof_of(fl, by, sec, a.precision)
                ent = traffic_tab.get(k) if traffic_tab else None
                pb = ent["hbm_bytes_per_launch"] if ent else df
                bound, ach, peak, unit, frac = physical_roof(fl, pb, sec, a.precision)
                d = {"ms": sec * 1e3, "tflops": fl / sec / 1e12, "bound": bound, "frac": frac, "achieved": ach, "unit": unit,
                     "physical_bytes": pb, "physical_bytes_source": "pmc" if ent else "dataflow", "physical_gbs": pb / sec / 1e9,
                     "frac_8d_convention": d8["frac"], "bound_8d_convention": d8["bound"], "gbs_8d_convention": d8["gbs"],
                     "ai_8d": d8["ai"]}
                if pk_f is not None:
                    d["frac_of_measured"] = ach / (pk_f if bound == "mfma" else pk_b)
                layer_roof[k] = {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in d.items()}
        conv_fl = sum(costs[k][0] for k in per_step if k in costs)
        conv_by = sum(costs[k][1] for k in per_step if k in costs)
        step_s = elapsed / a.steps
        step_roof = {"conv_flops_per_step": conv_fl, "algorithmic_bytes_per_step": conv_by,
                     "gflop_per_clip": conv_fl / B / 1e9, "mbyte_per_clip": conv_by / B / 1e6,
                     "tflops": conv_fl / step_s / 1e12, "frac_of_mfma_peak": conv_fl / step_s / (PEAK_MFMA_TFLOPS[a.precision] * 1e12),
                     "gbs": conv_by / step_s / 1e9, "frac_of_hbm_peak": conv_by / step_s / (PEAK_HBM_GBS * 1e9),
                     "note": "per rank (one GPU); SURVEY 8(d) algorithmic figures over the driver-timed ms_per_step"}
        if pk_f is not None:
            step_roof.update(frac_of_measured_mfma_peak=conv_fl / step_s / (pk_f * 1e12), frac_of_measured_hbm_peak=conv_by / step_s / (pk_b * 1e9))
        breakdown = {k: {"n": n, "ms_total": round(t, 3)} for k, (n, t) in top}
        result = {
            "metric": "SED train clips/sec (60s,64-mel,9-layer CNN)", "value": value, "unit": "clips/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.precision,
            "data": "synthetic",
            "config": {"workload": f"Cnn_AvgPooling {a.config} widths {[c for c, _ in cfg]}, "
                                   + (f"{T}-frame log-mel crops ({fcfg.mel_bins} mel), " if a.frames else
                                      f"{a.seconds:g} s / {fcfg.working_sample_rate // 1000} kHz / {fcfg.mel_bins}-mel clips (T={T} frames, "
                                      f"NFFT {fcfg.NFFT}, hop {fcfg.hop_size}), ")
                                   + f"batch {B}/GPU, train step = "
                                   f"{'features->' if a.no_frontend else 'waveform->log-mel->'}fwd->BCE->bwd->Adam-amsgrad"
                                   + (", HIP-graph replay" if a.graph else ""),
                       "frontend_constants": a.frontend, "graph_replay": bool(a.graph),
                       "global_batch": world * B, "frames": T, "parallelism": f"dp{world}",
                       "frontend_in_step": not a.no_frontend, "frontend_overlapped": bool(pf is not None), "frontend_overlap_mode": int(a.overlap_frontend),
                       "sync_bn": bool(a.sync_bn) and grouped, "rccl_group": bool(grouped and a.backend == "nccl"),
                       "collectives_executed": bool(trainer.reducer.enabled),
                       "grad_buckets": [list(k) for k, _, _ in trainer.flat.buckets],
                       # where a data-parallel step's communication goes (round 6): every bucket's all-reduce timed in isolation before
                       # the timed region, the time the compute stream is blocked on the collectives after the backward's last kernel
                       # (= exposed communication; instrumented pass), and the SyncBN collectives issued per step
                       "comm": None if not trainer.reducer.enabled else {
                           "backend": a.backend, "buckets": comm_iso[:-1], "flat_single_allreduce": comm_iso[-1],
                           "exposed_ms_per_step": exposed_ms, "exposed_share_of_step": exposed_ms / (elapsed / a.steps * 1e3),
                           "syncbn_collectives_per_step": bn_calls,
                           "how": "isolated: 20 synchronous all-reduces per bucket, HIP events on the compute stream, MAX over ranks; exposed: "
                                  "events around GradAllReducer.finish()'s waits (the compute stream idles there), rank 0, instrumented pass"}},
            "loss": loss_val, "roofline": roof, "layer_roofline": layer_roof, "step_roofline": step_roof,
            "kernel_breakdown_ms": breakdown,
            "gpu_time_ms_per_step_sum_of_kernels": sum(per_step.values()),
        }
        # ---- CPU baseline (oracle = "port"), N=1 only ----------------------------------------
        if world == 1 and not a.no_cpu_baseline:
            from oracle import cnn_oracle as O
            from oracle import frontend_oracle as FO
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            Bc = B if B * T <= 2 * 6001 else 2           # (the reference-shape lines fit whole batches into the CPU budget)
            sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
            stepper = O.AutogradStepper(sd, cfg, 5.0, 1e-6)
            wave_c = wave[:Bc].cpu().numpy() if wave is not None else None
            y_c = y[:Bc].cpu()
            ocfg = FO.FrontEndConfig(fcfg.working_sample_rate, fcfg.frame_size, fcfg.hop_size, fcfg.NFFT)
            mean_c, std_c = (mean.cpu().numpy(), std.cpu().numpy()) if mean is not None else (None, None)

            def cpu_step():
                if a.no_frontend:
                    x = feats[:Bc].cpu()
                else:
                    lm = np.stack([FO.log_mel_from_waveform(wave_c[i][:, None], ocfg, mean_c, std_c)[0] for i in range(Bc)])
                    x = torch.from_numpy(lm[:, None].astype(np.float32))
                return stepper.step(x, y_c)

            # ATen's CPU conv scales poorly past the physical cores for this shape: probe a few thread
            # counts with one step each and keep the fastest (its count is what `cores` reports)
            best = None
            for nt in sorted({avail, 64, 32, 16, 8}):
                if nt > avail:
                    continue
                torch.set_num_threads(nt)
                cpu_step()
                tp = time.perf_counter()
                cpu_step()
                dtp = time.perf_counter() - tp
                if best is None or dtp < best[1]:
                    best = (nt, dtp)
                if dtp > 20:
                    break
            ncores = best[0]
            torch.set_num_threads(ncores)
            n, t1 = 0, time.perf_counter()
            while True:
                cpu_step()
                n += 1
                if time.perf_counter() - t1 > a.cpu_seconds or n >= 20:
                    break
            cpu_el = time.perf_counter() - t1
            result["cpu_baseline"] = {"value": Bc * n / cpu_el, "unit": "clips/s", "cores": ncores, "host_cores": avail,
                                      "threads_probed": "fastest of {8,16,32,64,all} torch threads; `cores` = threads used",
                                      "kind": "port",
                                      "sample": f"{n} train steps of batch {Bc} (same T={T}, same model/optimizer, "
                                                f"{'features' if a.no_frontend else 'numpy front-end + '}ATen autograd "
                                                f"restatement in oracle/), after 1 warm-up step"}
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(result) + "\n").encode())
        else:
            print(json.dumps(result))
    if grouped:
        dist.destroy_process_group()
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        os.close(json_fd)


if __name__ == "__main__":
    main()
