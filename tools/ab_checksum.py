#!/usr/bin/env python3
"""Prints checksums of one bf16 training step (logits, loss, every gradient) for A/B runs of kernel
variants selected by environment variables (e.g. SED_CONV_LDS_WEIGHTS=1)."""
import hashlib
import sys

import torch

sys.path.insert(0, ".")
import sed_amd  # noqa: E402

cfgs = {"main": [(32, 2), (64, 2), (128, 2), (128, 1)], "default": [(64, 2), (128, 2), (256, 2), (512, 1)]}
name = sys.argv[1] if len(sys.argv) > 1 else "main"
B, T = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4, 301)
torch.manual_seed(0)
model = sed_amd.Cnn_AvgPooling(1, cfgs[name], precision="bf16").cuda()
tr = sed_amd.FusedTrainer(model, lr=1e-3)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 1, T, 64, generator=g).cuda()
y = (torch.rand(B, T, 1, generator=g) > 0.8).float().cuda()
loss = tr.forward_backward(x, y)
plan = next(iter(model.engine._plans.values()))
out = model.engine.interpolate(plan)
torch.cuda.synchronize()


def h(t):
    return hashlib.sha1(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:12]


print("loss", float(loss), "logits", h(out), "finite", bool(torch.isfinite(tr.flat.g).all()))
for n in tr.flat.names:
    print(n, h(tr.flat.G[n]), float(tr.flat.G[n].norm()))
