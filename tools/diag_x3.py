#!/usr/bin/env python3
"""GPU box: where does a split-operand mode (f16x3 / bf16x3) leave the fp32 mode?  One train step of the main config on G2's main30 batch
in both precisions with the backward's stage snapshots; prints the relative L2 distance of every stored forward tensor, BatchNorm
coefficient, backward stage and parameter gradient, then every ReLU-backward decision the two modes take differently (g1 of a block is
gated with bn1(z1) > 0: a pre-activation within rounding of 0 can fall on either side) with the pre-activation both modes saw.
usage: python tools/diag_x3.py [f16x3|bf16x3] [main30|main13]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sed = importlib.import_module("soundeventdetection-pytorch_amd")
MAIN = [(32, 2), (64, 2), (128, 2), (128, 1)]
g = np.load(os.path.join(ROOT, "tests", "golden", "g2_train_steps.npz"))
mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
tag = sys.argv[2] if len(sys.argv) > 2 else "main30"


def run(prec):
    torch.manual_seed(0)
    m = sed.Cnn_AvgPooling(1, MAIN, precision=prec).cuda()
    x, y = torch.from_numpy(g[f"{tag}.x"]).cuda(), torch.from_numpy(g[f"{tag}.y"]).cuda()
    tr = sed.FusedTrainer(m, lr=1e-3, recall_factor=5.0)
    eng = m.engine
    P = tr.flat.tensor_dict()
    m.train()
    plan = eng.forward(x, P, training=True)
    eng.loss_and_grad(plan, y, 5.0)
    dbg = {}
    eng.backward(plan, P, tr.flat.G, debug=dbg)
    torch.cuda.synchronize()
    out = {}
    for bi, blk in enumerate(plan.layers):
        for j, ly in enumerate(blk):
            out[f"fwd z b{bi}c{j + 1}"] = ly.z.float().clone()
            out[f"fwd scale b{bi}c{j + 1}"] = ly.scale.clone()
            out[f"fwd shift b{bi}c{j + 1}"] = ly.shift.clone()
            out[f"bwd coef b{bi}c{j + 1}"] = ly.coef.clone()
        out[f"fwd y b{bi}"] = plan.y[bi].float().clone()
    out["logits"] = plan.pre.clone()
    for k, v in dbg.items():
        out["bwd " + k] = v
    for n in tr.flat.names:
        out["grad " + n] = tr.flat.G[n].clone()
    return out


a, b = run("fp32"), run(mode)
for k in a:
    d = (a[k].double() - b[k].double()).norm() / max(a[k].double().norm().item(), 1e-30)
    mx = (a[k].double() - b[k].double()).abs().max().item()
    print(f"{k:44s} rel-L2 {d.item():.3e}   max|d| {mx:.3e}   max|ref| {a[k].abs().max().item():.3e}")

print("\nReLU-backward decisions taken differently (g1 = relu'(bn1(z1)) * conv2^T(dz2)):")
for bi in range(4):
    ga, gb = a[f"bwd g1_{bi}"], b[f"bwd g1_{bi}"]
    C = a[f"fwd scale b{bi}c1"].numel()
    pa = a[f"fwd z b{bi}c1"] * a[f"fwd scale b{bi}c1"] + a[f"fwd shift b{bi}c1"]
    pb = b[f"fwd z b{bi}c1"] * b[f"fwd scale b{bi}c1"] + b[f"fwd shift b{bi}c1"]
    diff = (ga == 0) != (gb == 0)
    idx = diff.nonzero()
    print(f"  block {bi}: {int(diff.sum())} of {ga.numel()} decisions differ")
    for i in idx[:8].tolist():
        t = tuple(i)
        print(f"    element {t}: bn1(z1) fp32 mode {pa[t].item():+.3e}, {mode} {pb[t].item():+.3e};  g1 fp32 mode {ga[t].item():+.3e}, {mode} {gb[t].item():+.3e}")
