#!/usr/bin/env python3
"""Train-step throughput of the other model families of SURVEY 8(f) / BASELINE.json configs on one MI355X (GPU box):
the CRNN (Cnn_9 + biGRU-256, 60 s clips), the default-width CNN (64/128/256/512) and the raw-waveform M5 (24 kHz frames).
Synthetic inputs, bf16, FusedTrainer.train_step (forward + BCE + backward + Adam-amsgrad), features resident in HBM.
usage: python tools/bench_models.py [steps]"""
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
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
MAIN = [(32, 2), (64, 2), (128, 2), (128, 1)]
DEFAULT = [(64, 2), (128, 2), (256, 2), (512, 1)]


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
