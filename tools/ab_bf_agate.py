#!/usr/bin/env python3
"""Block 1's conv2 fused backward (64 -> 64 @ 3000 x 32, B = 32) with the ReLU gate of the data gradient taken from the activation tile
in LDS (SED_BF_AGATE=1, round 5 default) against the round-4 epilogue (SED_BF_AGATE=0), interleaved in one process; outputs compared.
usage: ab_bf_agate.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, ".")
import sed_amd  # noqa: E402

L = sed_amd._lib
lib = L.lib()
P = L.ptr
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev, bf = "cuda", torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
B, H, W, C = 32, 3000, 32, 64
g = torch.Generator(device=dev).manual_seed(11)
d = {}
d["x"] = torch.randn(B, H, W, C, device=dev, generator=g).to(bf)
d["z"] = torch.randn(B, H, W, C, device=dev, generator=g).to(bf)
d["dy"] = torch.randn(B, H // 2, W // 2, C, device=dev, generator=g).to(bf)
d["sc_i"], d["sh_i"] = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.3
d["mu_i"], d["is_i"] = torch.randn(C, device=dev, generator=g) * 0.1, torch.rand(C, device=dev, generator=g) + 0.5
d["sc_o"], d["sh_o"] = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.1
d["ca"], d["cb"], d["cc"] = torch.randn(C, device=dev, generator=g), torch.randn(C, device=dev, generator=g) * 0.1, torch.randn(C, device=dev, generator=g) * 0.1
w = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05
d["wt"] = torch.empty(9 * C * C, device=dev, dtype=bf)
L.check(lib.sed_pack_conv_weight(1, P(w), P(d["wt"]), C, C, C, C, 1, st))
d["dwp"] = torch.empty(9 * C * C, device=dev)
d["dw"] = torch.empty(C, C, 3, 3, device=dev)
d["ws"] = torch.empty(lib.sed_conv_wgrad_ws_floats(B, H, W, C, C), device=dev)
d["np"] = lib.sed_conv_nparts(B, H, W)
d["part"] = torch.empty(d["np"], 2, C, device=dev)
d["dx"] = torch.empty(B, H, W, C, device=dev, dtype=bf)


def run(agate):
    os.environ["SED_BF_AGATE"] = agate
    lib.sed_config_reload()
    L.check(lib.sed_conv3x3_bwd_fused(1, 1, P(d["x"]), P(d["sc_i"]), P(d["sh_i"]), 1, P(d["dy"]), P(d["z"]), P(d["sc_o"]), P(d["sh_o"]), P(d["ca"]),
                                      P(d["cb"]), P(d["cc"]), 2, P(d["wt"]), P(d["dx"]), 2, P(d["x"]), None, P(d["sc_i"]), P(d["sh_i"]), P(d["mu_i"]),
                                      P(d["is_i"]), P(d["part"]), d["np"], None, P(d["dwp"]), P(d["ws"]), B, H, W, C, C, P(d["dw"]), C, C, st))


def timeit(agate, iters=10):
    run(agate)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run(agate)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


run("0")
torch.cuda.synchronize()
ref = (d["dx"].clone(), d["part"].sum(0).clone(), d["dw"].clone())
run("1")
torch.cuda.synchronize()
ps = d["part"].sum(0)
print("activation-tile gate vs z1 gate: dx identical", torch.equal(d["dx"], ref[0]), " dW identical", torch.equal(d["dw"], ref[2]),
      " BN1 sums max rel diff %.2e" % float(((ps - ref[1]).abs() / ref[1].abs().clamp_min(1e-6)).max()))
res = {}
for r in range(rounds):
    for a in ("0", "1"):
        res.setdefault(a, []).append(timeit(a))
for k, v in res.items():
    v = sorted(v)
    print(f"b1c2 fused backward (+ wgrad reduce) SED_BF_AGATE={k}   median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}")
