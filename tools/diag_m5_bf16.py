#!/usr/bin/env python3
"""M5 bf16 engine against oracle/m5_oracle_bf16.py (test infrastructure; this tool is a test aid): per-parameter gradient cosine /
norm ratio at the test's 64 frames and at more frames (noise floor or systematic?).  usage: diag_m5_bf16.py [frames ...]"""
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import m5_oracle as M            # noqa: E402
from oracle import m5_oracle_bf16 as MB      # noqa: E402
sed = importlib.import_module("soundeventdetection-pytorch_amd")
g7 = np.load("tests/golden/g7_m5.npz")
sd = {k[4:]: torch.from_numpy(g7[k]) for k in g7.files if k.startswith("sd0.")}
L_ = int(sys.argv[2]) if len(sys.argv) > 2 else 31680
for nf in [int(sys.argv[1])] if len(sys.argv) > 1 else [64]:
    gen = torch.Generator().manual_seed(64)
    x = 0.1 * torch.randn(nf, 1, L_, generator=gen)
    y = (torch.rand(nf, generator=gen) > 0.7).float()
    x[y > 0] += 0.2 * torch.sin(torch.arange(L_) * 0.05)
    m = sed.M5(1, precision="bf16")
    m.load_state_dict(sd)
    m.to("cuda:0").train()
    out = m(x.cuda())
    loss = sed.WeightedBCE(5, False)(out, y.cuda())
    loss.backward()
    loss_b, logits_b, grads_b, _ = MB.train_step_grads_bf16(x, y, sd, 5.0)
    loss_o, logits_o, grads_o, _ = M.train_step_grads(x, y, sd, 5.0)
    print(f"frames {nf} L {L_}: loss engine {loss.item():.6f} bf16-oracle {float(loss_b):.6f} fp32-oracle {float(loss_o):.6f}")
    for n, p in m.named_parameters():
        a = p.grad.double().cpu().flatten()
        b, c = grads_b[n].double().flatten(), grads_o[n].double().flatten()
        if float(b.norm()) < 1e-12:
            continue
        cos = lambda u, v: float((u @ v) / (u.norm() * v.norm() + 1e-30))
        print(f"  {n:24s} engine~bf16oracle cos {cos(a, b):.6f} ratio {float(a.norm() / b.norm()):.4f} | engine~fp32 {cos(a, c):.4f} | bf16oracle~fp32 {cos(b, c):.4f}")
