#!/usr/bin/env python3
"""In-process A/B of the wide weight-gradient kernel (csrc/sed_wgrad_wide.hip, round 5) against conv_wgrad3_kernel (csrc/sed_wgrad.hip,
SED_WGRAD_WIDE=0) on the weight-gradient launches of blocks 2-3 at the bench geometry (B = 32) and of the class-default widths (B = 16),
interleaved rounds on ONE device; each timing covers the launch + its slab reduction (what the step pays).
usage: ab_wgrad_wide.py [rounds] [bench|default|all]"""
import os
import sys

import torch

sys.path.insert(0, ".")
import sed_amd  # noqa: E402

L = sed_amd._lib
lib = L.lib()
P = L.ptr
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
which = sys.argv[2] if len(sys.argv) > 2 else "all"
dev, bf = "cuda", torch.bfloat16
st = torch.cuda.current_stream().cuda_stream


def mk(B, H, W, Cin, Cout, pool, conv2):
    d = dict(B=B, H=H, W=W, Cin=Cin, Cout=Cout, pool=pool, conv2=conv2)
    d["x"] = torch.randn(B, H, W, Cin, device=dev).to(bf)
    d["z"] = torch.randn(B, H, W, Cout, device=dev).to(bf)
    d["g"] = torch.randn(B, H, W, Cout, device=dev).to(bf)
    d["dy"] = torch.randn(B, H // pool, W // pool, Cout, device=dev).to(bf)
    d["sc_i"], d["sh_i"] = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    d["sc_o"], d["sh_o"] = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
    d["ca"], d["cb"], d["cc"] = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev) * 0.1, torch.randn(Cout, device=dev) * 0.1
    d["dwp"] = torch.empty(9 * Cin * Cout, device=dev)
    d["dw"] = torch.empty(Cout, Cin, 3, 3, device=dev)
    ws = 0
    for k in ("0", "1"):
        os.environ["SED_WGRAD_WIDE"] = k
        lib.sed_config_reload()
        ws = max(ws, lib.sed_conv_wgrad_ws_floats(B, H, W, Cin, Cout))
    d["ws"] = torch.empty(ws, device=dev)
    d["dz"] = torch.empty(B, H, W, Cout, device=dev, dtype=bf)
    return d


def run(d):
    B, H, W, Ci, Co = d["B"], d["H"], d["W"], d["Cin"], d["Cout"]
    if d["conv2"]:
        L.check(lib.sed_conv3x3_wgrad_fused_u(1, 1, P(d["x"]), P(d["sc_i"]), P(d["sh_i"]), 1, P(d["dy"]), P(d["z"]), P(d["sc_o"]), P(d["sh_o"]),
                                              P(d["ca"]), P(d["cb"]), P(d["cc"]), d["pool"], P(d["dz"]), P(d["dwp"]), P(d["ws"]), B, H, W, Ci, Co,
                                              P(d["dw"]), Co, Ci, st))
    else:
        L.check(lib.sed_conv3x3_wgrad_fused_u(1, 0, P(d["x"]), None, None, 2, P(d["g"]), P(d["z"]), None, None, P(d["ca"]), P(d["cb"]), P(d["cc"]), 1,
                                              P(d["dz"]), P(d["dwp"]), P(d["ws"]), B, H, W, Ci, Co, P(d["dw"]), Co, Ci, st))


def timeit(d, iters=10):
    run(d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run(d)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


layers = []
if which in ("bench", "all"):
    layers += [("bench b2c1 64->128 1500x16", mk(32, 1500, 16, 64, 128, 2, False)),
               ("bench b2c2 128->128 1500x16 pool2", mk(32, 1500, 16, 128, 128, 2, True)),
               ("bench b3c1 128->128 750x8", mk(32, 750, 8, 128, 128, 1, False)),
               ("bench b3c2 128->128 750x8 pool1", mk(32, 750, 8, 128, 128, 1, True))]
if which in ("default", "all"):
    layers += [("default b1c1 64->128 3000x32", mk(16, 3000, 32, 64, 128, 2, False)),
               ("default b1c2 128->128 3000x32 pool2", mk(16, 3000, 32, 128, 128, 2, True)),
               ("default b2c1 128->256 1500x16", mk(16, 1500, 16, 128, 256, 2, False)),
               ("default b2c2 256->256 1500x16 pool2", mk(16, 1500, 16, 256, 256, 2, True)),
               ("default b3c1 256->512 750x8", mk(16, 750, 8, 256, 512, 1, False)),
               ("default b3c2 512->512 750x8 pool1", mk(16, 750, 8, 512, 512, 1, True))]
res = {}
check = {}
for r in range(rounds):
    for name, d in layers:
        for k, tag in (("0", "narrow (sed_wgrad.hip)"), ("1", "wide")):
            os.environ["SED_WGRAD_WIDE"] = k
            lib.sed_config_reload()
            res.setdefault((name, tag), []).append(timeit(d))
            if r == 0:
                check[(name, tag)] = (d["dw"].clone(), d["dz"].clone())
for name, d in layers:
    a, b = check[(name, "narrow (sed_wgrad.hip)")], check[(name, "wide")]
    rel = float((a[0] - b[0]).abs().max() / a[0].abs().max())
    print(f"{name:40s} dW wide vs narrow: max rel {rel:.2e}   dz identical: {bool(torch.equal(a[1], b[1]))}")
tot = {}
for (name, tag), v in res.items():
    v = sorted(v)
    d = dict(layers)[name]
    fl = 2.0 * 9 * d["Cin"] * d["Cout"] * d["B"] * d["H"] * d["W"]
    print(f"{name:40s} {tag:24s} median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}   {fl / v[len(v) // 2] * 1e-9:7.1f} TFLOP/s = {fl / v[len(v) // 2] * 1e-9 / 2500:.3f} of 2.5 PF")
    key = (name.split()[0], tag)
    tot[key] = tot.get(key, 0.0) + v[len(v) // 2]
print("sum of medians:", {f"{k[0]} {k[1]}": round(v, 4) for k, v in tot.items()})
