#!/usr/bin/env python3
"""In-process A/B of the backward of blocks 2-3 at the bench geometry (B = 32): the two-kernel form (weight gradient produces and
writes dz, the data-gradient launch reads it back) against the cin-sliced fused kernel (csrc/sed_bwd_fused_cs.hip: dz only in
LDS), interleaved rounds on ONE device.   usage: ab_fused_cs.py [rounds]"""
import sys

import torch

sys.path.insert(0, ".")
import sed_amd  # noqa: E402

L = sed_amd._lib
lib = L.lib()
P = L.ptr
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev, bf = "cuda", torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
B = 32


def mk(H, W, Cin, Cout, pool):
    d = dict(H=H, W=W, Cin=Cin, Cout=Cout, pool=pool)
    d["x"] = torch.randn(B, H, W, Cin, device=dev).abs().to(bf)
    d["z"] = torch.randn(B, H, W, Cout, device=dev).to(bf)
    d["g"] = torch.randn(B, H, W, Cout, device=dev).to(bf)
    d["dy"] = torch.randn(B, H // pool, W // pool, Cout, device=dev).to(bf)
    d["cnt"] = torch.randint(0, 5, (B, H, W, Cin), device=dev, dtype=torch.int32).to(torch.uint8)
    d["sc_i"], d["sh_i"] = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    d["mu_i"], d["is_i"] = torch.randn(Cin, device=dev) * 0.1, torch.rand(Cin, device=dev) + 0.5
    d["sc_o"], d["sh_o"] = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
    d["ca"], d["cb"], d["cc"] = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev) * 0.1, torch.randn(Cout, device=dev) * 0.1
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    d["wt"] = torch.empty(9 * Cin * Cout, device=dev, dtype=bf)
    L.check(lib.sed_pack_conv_weight(1, P(w), P(d["wt"]), Cout, Cin, Cout, Cin, 1, st))
    d["dwp"] = torch.empty(9 * Cin * Cout, device=dev)
    d["dw"] = torch.empty(Cout, Cin, 3, 3, device=dev)
    d["ws"] = torch.empty(lib.sed_conv_wgrad_ws_floats(B, H, W, Cin, Cout), device=dev)
    d["np"] = lib.sed_conv_nparts(B, H, W)
    d["part"] = torch.empty(d["np"] * 2 * max(Cin, Cout), device=dev)
    d["dz"] = torch.empty(B, H, W, Cout, device=dev, dtype=bf)
    d["dx"] = torch.empty(B, H, W, Cin, device=dev, dtype=bf)
    d["flag"] = torch.zeros(1, device=dev, dtype=torch.int32)
    return d


def c1_unfused(d):
    H, W, Ci, Co = d["H"], d["W"], d["Cin"], d["Cout"]
    L.check(lib.sed_conv3x3_wgrad_fused_u(1, 0, P(d["x"]), None, None, 2, P(d["g"]), P(d["z"]), None, None, P(d["ca"]), P(d["cb"]), P(d["cc"]), 1,
                                          P(d["dz"]), P(d["dwp"]), P(d["ws"]), B, H, W, Ci, Co, P(d["dw"]), Co, Ci, st))
    L.check(lib.sed_conv3x3_dgrad_poolstats(1, P(d["dz"]), P(d["wt"]), P(d["dx"]), P(d["x"]), P(d["cnt"]), P(d["sc_i"]), P(d["sh_i"]), P(d["mu_i"]),
                                            P(d["is_i"]), P(d["part"]), d["np"], P(d["flag"]), B, H, W, Co, Ci, st))


def c1_fused(d):
    H, W, Ci, Co = d["H"], d["W"], d["Cin"], d["Cout"]
    L.check(lib.sed_conv3x3_bwd_fused(1, 0, P(d["x"]), None, None, 2, P(d["g"]), P(d["z"]), None, None, P(d["ca"]), P(d["cb"]), P(d["cc"]), 1,
                                      P(d["wt"]), P(d["dx"]), 4, P(d["x"]), P(d["cnt"]), P(d["sc_i"]), P(d["sh_i"]), P(d["mu_i"]), P(d["is_i"]),
                                      P(d["part"]), d["np"], P(d["flag"]), P(d["dwp"]), P(d["ws"]), B, H, W, Ci, Co, P(d["dw"]), Co, Ci, st))


def c2_unfused(d):
    H, W, Ci, Co, pool = d["H"], d["W"], d["Cin"], d["Cout"], d["pool"]
    L.check(lib.sed_conv3x3_wgrad_fused_u(1, 1, P(d["x"]), P(d["sc_i"]), P(d["sh_i"]), 1, P(d["dy"]), P(d["z"]), P(d["sc_o"]), P(d["sh_o"]), P(d["ca"]),
                                          P(d["cb"]), P(d["cc"]), pool, P(d["dz"]), P(d["dwp"]), P(d["ws"]), B, H, W, Ci, Co, P(d["dw"]), Co, Ci, st))
    L.check(lib.sed_conv3x3_fwd(1, 0, 2, P(d["dz"]), None, None, P(d["wt"]), P(d["dx"]), P(d["x"]), P(d["sc_i"]), P(d["sh_i"]), P(d["mu_i"]), P(d["is_i"]),
                                P(d["part"]), B, H, W, Co, Ci, st))


def c2_fused(d):
    H, W, Ci, Co, pool = d["H"], d["W"], d["Cin"], d["Cout"], d["pool"]
    L.check(lib.sed_conv3x3_bwd_fused(1, 1, P(d["x"]), P(d["sc_i"]), P(d["sh_i"]), 1, P(d["dy"]), P(d["z"]), P(d["sc_o"]), P(d["sh_o"]), P(d["ca"]),
                                      P(d["cb"]), P(d["cc"]), pool, P(d["wt"]), P(d["dx"]), 2, P(d["x"]), None, P(d["sc_i"]), P(d["sh_i"]), P(d["mu_i"]),
                                      P(d["is_i"]), P(d["part"]), d["np"], None, P(d["dwp"]), P(d["ws"]), B, H, W, Ci, Co, P(d["dw"]), Co, Ci, st))


def timeit(fn, d, iters=10):
    fn(d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn(d)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


layers = [("b2c1 64->128 1500x16", mk(1500, 16, 64, 128, 2), c1_unfused, c1_fused),
          ("b2c2 128->128 1500x16", mk(1500, 16, 128, 128, 2), c2_unfused, c2_fused),
          ("b3c1 128->128 750x8", mk(750, 8, 128, 128, 1), c1_unfused, c1_fused),
          ("b3c2 128->128 750x8 pool1", mk(750, 8, 128, 128, 1), c2_unfused, c2_fused)]
res = {}
for r in range(rounds):
    for name, d, fu, ff in layers:
        res.setdefault(name + " two kernels", []).append(timeit(fu, d))
        res.setdefault(name + " fused", []).append(timeit(ff, d))
tot = {"two kernels": 0.0, "fused": 0.0}
for k, v in res.items():
    v = sorted(v)
    print(f"{k:40s} median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}")
    tot["fused" if k.endswith("fused") else "two kernels"] += v[len(v) // 2]
print("sum of medians:", {k: round(v, 4) for k, v in tot.items()})
