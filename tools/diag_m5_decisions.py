#!/usr/bin/env python3
"""M5 bf16 engine against oracle/m5_oracle_bf16.py with the branch decisions shared (test aid, round 5): per layer how many ReLU / arg-max
decisions differ between the engine (rebuilt from its stored z, scale, shift) and the oracle, how far from a tie those are, and the
gradient cosines with (a) no sharing, (b) sharing at near-ties (the test's rule), (c) ALL engine decisions taken.
usage: diag_m5_decisions.py [frames]"""
import importlib
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from oracle import m5_oracle as M            # noqa: E402
from oracle import m5_oracle_bf16 as MB      # noqa: E402
sed = importlib.import_module("soundeventdetection-pytorch_amd")
g7 = np.load("tests/golden/g7_m5.npz")
sd = {k[4:]: torch.from_numpy(g7[k]) for k in g7.files if k.startswith("sd0.")}
nf, L_ = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 31680
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
plan = next(iter(m.engine._plans.values()))
dec, zs = [], []
for ly in plan.layers:
    N_, H_, _, C_ = ly.z.shape
    z_e = ly.z.float().permute(0, 2, 3, 1).reshape(N_ * 8, C_, H_).cpu()
    pre_e = torch.addcmul(ly.shift.cpu()[None, :, None], z_e, ly.scale.cpu()[None, :, None])
    e = {"mask": pre_e > 0, "idx": None}
    if ly.pool:
        e["idx"] = F.max_pool1d(torch.relu(pre_e), 4, 4, return_indices=True)[1]
    dec.append(e)
    zs.append((z_e, ly.scale.cpu(), ly.shift.cpu(), pre_e))

# the oracle's own forward, layer by layer, to compare z / decisions
P = {k: v.double() for k, v in sd.items()}
a = MB.round_bf16(x.double())
print(f"{'layer':18s} {'z differs':>10s} {'|dz|/ulp max':>12s} {'scale rel':>10s} {'relu differ':>11s} {'far (>tie)':>11s} {'argmax differ':>13s} {'far':>8s}")
for li, (conv, bn, cin, cout, k, s, p, pool) in enumerate(M.layer_list()):
    z = MB.round_bf16(F.conv1d(a, MB.round_bf16(P[conv + ".weight"]), None, stride=s, padding=p))
    co = MB._bn_coeffs(z, P[bn + ".weight"], P[bn + ".bias"])
    pre = z * co["scale"][None, :, None] + co["shift"][None, :, None]
    z_e, sc_e, sh_e, pre_e = zs[li]
    ulp = MB._bf16_ulp(z)
    dzu = ((z_e.double() - z).abs() / ulp)
    tie = MB.tie_tolerance(z, co["scale"])
    mask = pre > 0
    dm = dec[li]["mask"] != mask
    far = dm & (pre.abs() > tie)
    act = torch.relu(pre)
    line = f"{conv:18s} {int((dzu > 0).sum()):10d} {float(dzu.max()):12.2f} {float(((sc_e.double() - co['scale']).abs() / co['scale'].abs()).max()):10.2e} {int(dm.sum()):11d} {int(far.sum()):11d}"
    if pool:
        yp, idx = F.max_pool1d(act, 4, 4, return_indices=True)
        di = dec[li]["idx"] != idx
        at_e = act.gather(2, dec[li]["idx"])
        farI = di & (at_e < yp - tie.gather(2, idx))
        line += f" {int(di.sum()):13d} {int(farI.sum()):8d}"
        a = MB.round_bf16(yp)
    else:
        a = MB.round_bf16(act)
    print(line)


def report(tag, grads):
    rows = []
    for n, p_ in m.named_parameters():
        b = grads[n].double().flatten()
        if float(b.norm()) < 1e-12:
            continue
        a_ = p_.grad.double().cpu().flatten()
        rows.append((n, float((a_ @ b) / (a_.norm() * b.norm() + 1e-30)), float(a_.norm() / b.norm())))
    worst = min(rows, key=lambda r: r[1])
    print(f"{tag:34s} worst cosine {worst[1]:.6f} ({worst[0]}), " + " ".join(f"{r[0].split('.')[0][-1]}.{r[0].split('.')[1]}:{r[1]:.4f}" for r in rows if r[0].endswith("weight") and "conv" in r[0] and r[0].split(".")[1] in ("0", "3")))


report("(a) no sharing", MB.train_step_grads_bf16(x, y, sd, 5.0)[2])
st = {}
report("(b) sharing at near-ties", MB.train_step_grads_bf16(x, y, sd, 5.0, take_decisions=dec, decision_stats=st)[2])
print("    borrowed:", {k: v[0] for k, v in st.items() if v[0]})
import oracle.m5_oracle_bf16 as MBm
orig = MBm._bf16_ulp
MBm._bf16_ulp = lambda v: torch.full_like(v, 1e30)          # every decision counts as a near-tie: ALL engine decisions are taken
st = {}
report("(c) ALL engine decisions", MB.train_step_grads_bf16(x, y, sd, 5.0, take_decisions=dec, decision_stats=st)[2])
print("    borrowed:", {k: v[0] for k, v in st.items() if v[0]})
MBm._bf16_ulp = orig
