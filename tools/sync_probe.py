#!/usr/bin/env python3
"""Does the SyncBN reduction path (row sums -> one row -> finalize) change the bf16 forward by itself?  (GPU box)
Single process; a world-1 stand-in for the process group makes the all-reduce a no-op."""
import importlib
import sys

import torch

sys.path.insert(0, ".")
sed = importlib.import_module("soundeventdetection-pytorch_amd")
CFG = [(32, 2), (64, 2), (128, 2), (128, 1)]
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"


class One:
    world = 1

    def all_reduce(self, t):
        return t


def run(sync):
    torch.manual_seed(0)
    m = sed.Cnn_AvgPooling(1, CFG, precision=prec).cuda()
    g = torch.Generator().manual_seed(123)
    x = torch.randn(4, 1, 64, 64, generator=g).cuda()
    y = (torch.rand(4, 64, 1, generator=g) > 0.75).float().cuda()
    tr = sed.FusedTrainer(m, lr=1e-3, recall_factor=5.0)
    if sync:
        m.engine.bn_sync = One()
    loss = tr.forward_backward(x, y)
    plan = next(iter(m.engine._plans.values()))
    lg = m.engine.interpolate(plan).cpu()
    sc = [ly.scale.clone().cpu() for blk in plan.layers for ly in blk]
    sh = [ly.shift.clone().cpu() for blk in plan.layers for ly in blk]
    return lg, tr.flat.g.clone().cpu(), sc, sh


a, b = run(False), run(True)
print("logits max diff", float((a[0] - b[0]).abs().max()))
print("grad rel", float((a[1].double() - b[1].double()).norm() / a[1].double().norm()))
for i, (s0, s1, h0, h1) in enumerate(zip(a[2], b[2], a[3], b[3])):
    print(f"layer {i}: scale rel diff {float(((s0 - s1).abs() / (s0.abs() + 1e-12)).max()):.3e}  shift abs diff {float((h0 - h1).abs().max()):.3e}")
