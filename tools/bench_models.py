#!/usr/bin/env python3
"""Train-step throughput of the other model families of SURVEY 8(f) / BASELINE.json configs on one MI355X (GPU box):
the CRNN (Cnn_9 + biGRU-256, 60 s clips), the default-width CNN (64/128/256/512) and the raw-waveform M5 (24 kHz frames).
Synthetic inputs, bf16, FusedTrainer.train_step (forward + BCE + backward + Adam-amsgrad), features resident in HBM.
usage: python tools/bench_models.py [steps]
       python tools/bench_models.py --json crnn|m5 [--steps K --warmup W --batch B]    (round 5: one model, bench.py's JSON line schema)"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sed = importlib.import_module("soundeventdetection-pytorch_amd")
ms = importlib.import_module("soundeventdetection-pytorch_amd.models.spectogram_models")
mw = importlib.import_module("soundeventdetection-pytorch_amd.models.waveform_models")
MAIN = [(32, 2), (64, 2), (128, 2), (128, 1)]
DEFAULT = [(64, 2), (128, 2), (256, 2), (512, 1)]

# ---- round 5: `--json crnn | m5`: ONE model (BASELINE.json configs 4 / 5 at their per-GPU batch) in bench.py's line schema -------
# label prefix -> regex on the demangled kernel name: joins the engine's launch labels (HIP events) with the per-kernel-name PMC
# tables tools/profile_r05_models.sh writes (profiles/r05_<model>_hbm_by_kernel.json)
LABEL_KERNEL = {
    "crnn": {"sed_gru_seq_fwd": r"gru_seq_fwd16h_kernel", "sed_gru_seq_bwd": r"gru_seq_bwd16h_kernel",
             "sed_conv3x3_bwd_fused_c1": r"conv_bwd_fused_c1_kernel", "sed_conv3x3_fwd_c1": r"conv_pc_kernel<64, 32, 2, 1"},
    "m5": {"sed_m5_conv1_bn_relu_pool_fwd": r"m5_conv1_fwd_mfma_kernel<2>", "sed_m5_conv1_stats": r"m5_conv1_fwd_mfma_kernel<1>",
           "sed_m5_conv1_wgrad": r"m5_conv1_wgrad_mfma_kernel", "sed_maxpool4_pooled_stats": r"maxpool4_pooled_stats_kernel",
           "sed_conv3x3_wgrad_fused:bwd conv_block2.0": r"conv_wgrad3_kernel<8, 2, 2, 2, 0", "sed_conv3x3_wgrad_fused:bwd conv_block2.3": r"conv_wgrad3_kernel<8, 2, 2, 2, 1"},
}


def json_mode(argv):
    import argparse
    import json
    import re
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", choices=["crnn", "m5"], required=True)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="crnn: clips (default 16 = config 4's per-GPU batch); m5: frames (default 2880 = 64 clips of 60 s)")
    ap.add_argument("--no-timer-pass", action="store_true", help="skip the instrumented pass (runs under rocprofv3 --pmc keep it short)")
    a = ap.parse_args(argv)
    g = torch.Generator(device="cuda").manual_seed(0)
    torch.manual_seed(0)
    if a.json == "crnn":
        B, T = (a.batch or 16), 6001
        x = torch.randn(B, 1, T, 64, device="cuda", generator=g)
        y = (torch.rand(B, T, 1, device="cuda", generator=g) < 0.04).float()
        model = ms.Crnn_AvgPooling(1, MAIN, precision="bf16", gru_hidden=256).cuda()
        units, unit = B, "clips/s"
        metric = "SED train clips/sec (60s,64-mel, CRNN = Cnn_9 + biGRU-256)"
        workload = f"Crnn_AvgPooling main widths + biGRU-256, 60 s / 64-mel clips (T={T} frames, {T // 8} recurrence steps), batch {B}/GPU " \
                   f"(BASELINE config 4: 128 clips over 8 GPUs), train step = features->fwd->BCE->bwd->Adam-amsgrad"
    else:
        B = a.batch or 2880
        x = torch.randn(B, 1, 31680, device="cuda", generator=g) * 0.1
        y = (torch.rand(B, device="cuda", generator=g) < 0.1).float()
        model = mw.M5(1, precision="bf16").cuda()
        units, unit = B, "frames/s"
        metric = "SED train frames/sec (raw-waveform M5, 31680-sample frames @ 24 kHz)"
        workload = f"M5 (waveform_models.py), {B} frames of 31680 samples (= {B / 45:g} clips of 60 s @ 24 kHz; BASELINE config 5: batch 64 clips), " \
                   f"train step = waveform->fwd->BCE->bwd->Adam-amsgrad"
    tr = sed.FusedTrainer(model, lr=1e-6, recall_factor=5.0)
    for _ in range(a.warmup):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.train_step(x, y)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {"metric": metric, "value": units * a.steps / el, "unit": unit, "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
           "data": "synthetic", "config": {"workload": workload, "global_batch": B, "parallelism": "dp1"}, "loss": float(loss)}
    if not a.no_timer_pass:
        timer = sed.engine.KernelTimer()
        tr.engine.timer = timer
        for _ in range(a.steps):
            tr.train_step(x, y)
        torch.cuda.synchronize()
        tr.engine.timer = None
        summ = timer.summary()
        top = sorted(summ.items(), key=lambda kv: -kv[1][1])
        out["kernel_breakdown_ms"] = {k: {"n": n, "ms_total": round(t, 3), "ms_per_step": round(t / a.steps, 4)} for k, (n, t) in top}
        out["gpu_time_ms_per_step_sum_of_kernels"] = sum(t for _, (n, t) in top) / a.steps
        # per-label roofline entries for the labels the PMC table can be joined with (bytes per launch by kernel name)
        tab_path = os.path.join(ROOT, "profiles", f"r05_{a.json}_hbm_by_kernel.json")
        tab = json.load(open(tab_path)) if os.path.exists(tab_path) else {}
        lr = {}
        for k, (n, t) in top:
            rx = next((r for pre, r in LABEL_KERNEL[a.json].items() if k.startswith(pre)), None)
            ent = next((v for name, v in tab.items() if rx and re.search(rx, name)), None) if rx else None
            if ent:
                sec = t / n / 1e3
                lr[k] = {"ms": round(sec * 1e3, 4), "hbm_bytes_per_launch": ent["hbm_bytes_per_launch"], "gbs": round(ent["hbm_bytes_per_launch"] / sec / 1e9, 1),
                         "frac_of_hbm_peak": round(ent["hbm_bytes_per_launch"] / sec / 8e12, 4), "kernel": ent.get("kernel")}
        out["layer_roofline"] = lr
        dom, (dn, dt_) = top[0]
        roof = {"kernel": dom, "launches": dn, "avg_ms": dt_ / dn, "share_of_gpu_time": dt_ / sum(t for _, (n, t) in top),
                "timing": "HIP events around each launch on the launch stream, instrumented pass of the same steps after the timed region"}
        if dom in lr:
            roof.update({"bound": "hbm", "achieved": lr[dom]["gbs"], "peak": 8000.0, "unit": "GB/s", "frac": lr[dom]["frac_of_hbm_peak"],
                         "traffic": lr[dom]["hbm_bytes_per_launch"], "traffic_source": os.path.relpath(tab_path, ROOT)})
        else:
            roof.update({"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None})
        out["roofline"] = roof
        if a.json == "crnn":
            tsteps = T // 8
            f = next((t / n for k, (n, t) in top if k.startswith("sed_gru_seq_fwd")), None)
            b = next((t / n for k, (n, t) in top if k.startswith("sed_gru_seq_bwd")), None)
            wg = B * 2 // 2                       # (clip, direction) pairs / 2 rows per workgroup
            out["recurrence"] = {"steps": tsteps, "fwd_ms": f, "bwd_ms": b, "fwd_us_per_step": f * 1e3 / tsteps if f else None,
                                 "bwd_us_per_step": b * 1e3 / tsteps if b else None, "workgroups": wg, "cus_busy": min(256, wg),
                                 "cu_occupancy": min(256, wg) / 256.0,
                                 "note": "one 2-row chunk of one direction per workgroup, one workgroup per CU: B*2/2 CUs run the sequential chain"}
    print(json.dumps(out))


if "--json" in sys.argv:
    json_mode(sys.argv[1:])
    sys.exit(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def run(name, model, x, y, unit, per_step):
    model = model.cuda()
    tr = sed.FusedTrainer(model, lr=1e-6, recall_factor=5.0)
    for _ in range(3):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.train_step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name}: {dt * 1e3:8.3f} ms/step  {per_step / dt:10.1f} {unit}  loss {float(loss):.4f}", flush=True)


g = torch.Generator(device="cuda").manual_seed(0)
T = 6001
x = torch.randn(32, 1, T, 64, device="cuda", generator=g)
y = (torch.rand(32, T, 1, device="cuda", generator=g) < 0.04).float()
torch.manual_seed(0)
run("Cnn_AvgPooling main   B=32 T=6001", ms.Cnn_AvgPooling(1, MAIN, precision="bf16"), x, y, "clips/s", 32)
run("Crnn_AvgPooling main  B=32 T=6001 (biGRU-256)", ms.Crnn_AvgPooling(1, MAIN, precision="bf16", gru_hidden=256), x, y, "clips/s", 32)
x16, y16 = x[:16].contiguous(), y[:16].contiguous()
run("Crnn_AvgPooling main  B=16 T=6001 (biGRU-256; BASELINE config 4 = 16 clips per GPU)",
    ms.Crnn_AvgPooling(1, MAIN, precision="bf16", gru_hidden=256), x16, y16, "clips/s", 16)
run("Cnn_AvgPooling default B=16 T=6001 (64/128/256/512)", ms.Cnn_AvgPooling(1, DEFAULT, precision="bf16"), x16, y16, "clips/s", 16)
del x, y, x16, y16
torch.cuda.empty_cache()

# raw-waveform M5: one frame = 31680 samples (waveform_configs frame size); a 60 s / 24 kHz clip = 45 such frames
xf = torch.randn(2880, 1, 31680, device="cuda", generator=g) * 0.1          # 64 clips x 45 frames
yf = (torch.rand(2880, device="cuda", generator=g) < 0.1).float()
try:
    run("M5 bf16  B=2880 frames (= 64 clips of 60 s @ 24 kHz)", mw.M5(1, precision="bf16"), xf, yf, "frames/s", 2880)
except Exception as e:      # noqa: BLE001
    print("M5 run failed:", repr(e))

# the reference's own shapes: 48 kHz / hop 15840 -> a 60 s recording is T = 182 frames, training crops are T = 30,
# main.py's default batch is small; these steps are launch-bound (~90 launches)
torch.manual_seed(0)
for (Bn, Tn, prec) in ((4, 30, "fp32"), (4, 30, "bf16"), (32, 182, "bf16"), (128, 30, "bf16")):
    xs = torch.randn(Bn, 1, Tn, 64, device="cuda", generator=g)
    ys = (torch.rand(Bn, Tn, 1, device="cuda", generator=g) < 0.04).float()
    run(f"Cnn_AvgPooling main {prec} B={Bn} T={Tn} (reference-native frames)", ms.Cnn_AvgPooling(1, MAIN, precision=prec), xs, ys, "clips/s", Bn)
