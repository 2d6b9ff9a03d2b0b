#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP pipeline with the CPU oracle (diagnostic; GPU box only).
Prints one line per intermediate tensor: max|err|, scale of the reference, and a PASS/FAIL flag.

    python tools/gpu_stage_check.py [--precision fp32|bf16] [--config tiny|main] [--B 2] [--T 30]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sed_amd  # noqa: E402
from oracle import cnn_oracle as O  # noqa: E402

CFGS = {"tiny": [(4, 2), (8, 2), (8, 2), (8, 1)], "main": [(32, 2), (64, 2), (128, 2), (128, 1)],
        "default": [(64, 2), (128, 2), (256, 2), (512, 1)]}


def nhwc_to_nchw(t, C):
    return t[..., :C].permute(0, 3, 1, 2).contiguous().cpu()


def report(name, got, ref, tol):
    got = got.detach().float().cpu().double()
    ref = ref.detach().double()
    if got.shape != ref.shape:
        print(f"{name:28s} SHAPE MISMATCH {tuple(got.shape)} vs {tuple(ref.shape)}")
        return False
    err = (got - ref).abs().max().item() if ref.numel() else 0.0
    scale = max(ref.abs().max().item(), 1e-30) if ref.numel() else 1.0
    ok = err <= tol * max(1.0, scale) or err / scale <= tol
    print(f"{name:28s} max|err| {err:10.3e}  ref-scale {scale:10.3e}  rel {err / scale:9.2e}  {'PASS' if ok else 'FAIL'}")
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--config", default="tiny")
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--T", type=int, default=30)
    ap.add_argument("--K", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    cfg = CFGS[a.config]
    tol = 2e-4 if a.precision == "fp32" else 6e-2
    torch.manual_seed(a.seed)
    model = sed_amd.Cnn_AvgPooling(a.K, cfg, precision=a.precision)
    with torch.no_grad():
        for blk in model.conv_blocks:
            for bn in (blk.bn1, blk.bn2):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.uniform_(-0.3, 0.3)
        model.event_fc.bias.uniform_(-0.1, 0.1)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.randn(a.B, 1, a.T, 64)
    y = (torch.rand(a.B, a.T, a.K) > 0.7).float()
    # ---- oracle (fp32 CPU) -------------------------------------------------------------------
    loss_o, logits_o, grads_o, ns_o, cache_o = O.train_step_grads(x, y, sd, cfg, 5.0)
    # ---- HIP ---------------------------------------------------------------------------------
    model = model.cuda().train()
    P = model._tensor_dict()
    eng = model.engine
    plan = eng.forward(x.cuda(), P, training=True)
    torch.cuda.synchronize()
    allok = True
    for bi, (c, pool) in enumerate(cfg):
        co = cache_o["blocks"][bi]
        for j in (1, 2):
            ly = plan.layers[bi][j - 1]
            allok &= report(f"blk{bi}.z{j}", nhwc_to_nchw(ly.z, c), co[f"z{j}"], tol)
            allok &= report(f"blk{bi}.mean{j}", ly.mean[:c], co[f"mean{j}"], tol)
            allok &= report(f"blk{bi}.invstd{j}", ly.invstd[:c], co[f"invstd{j}"], tol)
            if ly.coutp > c:
                z_pad = ly.z[..., c:].float().abs().max().item()
                print(f"blk{bi}.z{j} padded-channel max {z_pad:.3e}")
        allok &= report(f"blk{bi}.out", nhwc_to_nchw(plan.y[bi], c), co["out"], tol)
        for j in (1, 2):
            allok &= report(f"blk{bi}.bn{j}.running_var", P[f"conv_blocks.{bi}.bn{j}.running_var"],
                            ns_o[f"conv_blocks.{bi}.bn{j}.running_var"], tol)
    allok &= report("pre", plan.pre, cache_o["head"]["pre"], tol)
    loss = eng.loss_and_grad(plan, y.cuda(), 5.0)
    allok &= report("loss", loss[0], loss_o, tol)
    dlog_o = O.weighted_bce_bwd(logits_o, y, 5.0)
    dpre_o = dlog_o.reshape(a.B, -1, eng.ratio, a.K).sum(2)
    allok &= report("dpre", plan.dpre, dpre_o, tol)
    G = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
    dbg = {}
    eng.backward(plan, P, G, debug=dbg)
    torch.cuda.synchronize()
    for bi in reversed(range(len(cfg))):
        c = cfg[bi][0]
        co = cache_o["blocks"][bi]
        allok &= report(f"blk{bi}.dz2", nhwc_to_nchw(dbg[f"dz2_{bi}"], c), co["dz2"], tol)
        allok &= report(f"blk{bi}.dz1", nhwc_to_nchw(dbg[f"dz1_{bi}"], c), co["dz1"], tol)
        for n in O.PARAM_SUFFIXES:
            allok &= report(f"grad blk{bi}.{n}", G[f"conv_blocks.{bi}.{n}"], grads_o[f"conv_blocks.{bi}.{n}"], tol * 5)
    allok &= report("grad event_fc.weight", G["event_fc.weight"], grads_o["event_fc.weight"], tol * 5)
    allok &= report("grad event_fc.bias", G["event_fc.bias"], grads_o["event_fc.bias"], tol * 5)
    # autograd path of the module API
    model.zero_grad()
    out = model(x.cuda())
    allok &= report("module logits", out, logits_o, tol)
    crit = sed_amd.WeightedBCE(5, True)
    l2 = crit(out, y.cuda())
    l2.backward()
    allok &= report("module loss", l2, loss_o, tol)
    allok &= report("module grad conv_blocks.0.conv1.weight", model.conv_blocks[0].conv1.weight.grad,
                    grads_o["conv_blocks.0.conv1.weight"], tol * 5)
    print("ALL PASS" if allok else "SOME FAILED")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
