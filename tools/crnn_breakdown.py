#!/usr/bin/env python3
"""Per-kernel HIP-event breakdown of one CRNN train step (GPU box).  usage: python tools/crnn_breakdown.py [B]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sed = importlib.import_module("soundeventdetection-pytorch_amd")
ms = importlib.import_module("soundeventdetection-pytorch_amd.models.spectogram_models")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, 1, 6001, 64, device="cuda", generator=g)
y = (torch.rand(B, 6001, 1, device="cuda", generator=g) < 0.04).float()
torch.manual_seed(0)
model = ms.Crnn_AvgPooling(1, [(32, 2), (64, 2), (128, 2), (128, 1)], precision="bf16", gru_hidden=256).cuda()
tr = sed.FusedTrainer(model, lr=1e-6, recall_factor=5.0)
for _ in range(3):
    tr.train_step(x, y)
torch.cuda.synchronize()
timer = sed.engine.KernelTimer()
tr.engine.timer = timer
for _ in range(5):
    tr.train_step(x, y)
torch.cuda.synchronize()
tr.engine.timer = None
rows = sorted(((t / n * (n / 5), n // 5, k) for k, (n, t) in timer.summary().items()), reverse=True)
tot = sum(r[0] for r in rows)
for ms_, n, k in rows[:int(os.environ.get("TOPN", "14"))]:
    print(f"{ms_:8.3f} ms  x{n}  {k}")
print(f"{tot:8.3f} ms total")
