#!/usr/bin/env python3
"""In-process A/B of the block-1 backward at the bench geometry (B = 32, 3000 x 32): the two-kernel form (weight gradient writes
dz, data gradient reads it back) against the fused kernel and its build variants (make EXPERIMENTS=1: SED_BF_VAR), interleaved
rounds on ONE device (devices differ by 5-15 %: never compare across boxes).   usage: ab_fused.py [rounds] [vars e.g. 3,0,1,2]"""
import os
import sys

import torch

sys.path.insert(0, ".")
import sed_amd  # noqa: E402

L = sed_amd._lib
lib = L.lib()
P = L.ptr
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["3"]
dev, bf = "cuda", torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
B, H, W = 32, 3000, 32


def mk(Cin, Cout):
    d = {}
    d["x"] = torch.randn(B, H, W, Cin, device=dev).abs().to(bf)
    d["z"] = torch.randn(B, H, W, Cout, device=dev).to(bf)
    d["g"] = torch.randn(B, H, W, Cout, device=dev).to(bf)
    d["dy"] = torch.randn(B, H // 2, W // 2, Cout, device=dev).to(bf)
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


c1, c2 = mk(32, 64), mk(64, 64)


def c1_unfused():
    d = c1
    L.check(lib.sed_conv3x3_wgrad_fused_u(1, 0, P(d["x"]), None, None, 2, P(d["g"]), P(d["z"]), None, None, P(d["ca"]), P(d["cb"]), P(d["cc"]), 1,
                                          P(d["dz"]), P(d["dwp"]), P(d["ws"]), B, H, W, 32, 64, P(d["dw"]), 64, 32, st))
    L.check(lib.sed_conv3x3_dgrad_poolstats(1, P(d["dz"]), P(d["wt"]), P(d["dx"]), P(d["x"]), P(d["cnt"]), P(d["sc_i"]), P(d["sh_i"]), P(d["mu_i"]),
                                            P(d["is_i"]), P(d["part"]), d["np"], P(d["flag"]), B, H, W, 64, 32, st))


def c1_fused():
    d = c1
    L.check(lib.sed_conv3x3_bwd_fused(1, 0, P(d["x"]), None, None, 2, P(d["g"]), P(d["z"]), None, None, P(d["ca"]), P(d["cb"]), P(d["cc"]), 1,
                                      P(d["wt"]), P(d["dx"]), 4, P(d["x"]), P(d["cnt"]), P(d["sc_i"]), P(d["sh_i"]), P(d["mu_i"]), P(d["is_i"]),
                                      P(d["part"]), d["np"], P(d["flag"]), P(d["dwp"]), P(d["ws"]), B, H, W, 32, 64, P(d["dw"]), 64, 32, st))


def c2_unfused():
    d = c2
    L.check(lib.sed_conv3x3_wgrad_fused_u(1, 1, P(d["x"]), P(d["sc_i"]), P(d["sh_i"]), 1, P(d["dy"]), P(d["z"]), P(d["sc_o"]), P(d["sh_o"]), P(d["ca"]),
                                          P(d["cb"]), P(d["cc"]), 2, P(d["dz"]), P(d["dwp"]), P(d["ws"]), B, H, W, 64, 64, P(d["dw"]), 64, 64, st))
    L.check(lib.sed_conv3x3_fwd(1, 0, 2, P(d["dz"]), None, None, P(d["wt"]), P(d["dx"]), P(d["x"]), P(d["sc_i"]), P(d["sh_i"]), P(d["mu_i"]), P(d["is_i"]),
                                P(d["part"]), B, H, W, 64, 64, st))


def c2_fused():
    d = c2
    L.check(lib.sed_conv3x3_bwd_fused(1, 1, P(d["x"]), P(d["sc_i"]), P(d["sh_i"]), 1, P(d["dy"]), P(d["z"]), P(d["sc_o"]), P(d["sh_o"]), P(d["ca"]),
                                      P(d["cb"]), P(d["cc"]), 2, P(d["wt"]), P(d["dx"]), 2, P(d["x"]), None, P(d["sc_i"]), P(d["sh_i"]), P(d["mu_i"]),
                                      P(d["is_i"]), P(d["part"]), d["np"], None, P(d["dwp"]), P(d["ws"]), B, H, W, 64, 64, P(d["dw"]), 64, 64, st))


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


res = {}
for r in range(rounds):
    for name, fn in (("c1 unfused", c1_unfused), ("c2 unfused", c2_unfused)):
        res.setdefault(name, []).append(timeit(fn))
    for v in variants:
        os.environ["SED_BF_VAR"] = v.split("p")[0].split("a")[0]
        os.environ["SED_BF_PRIO"] = v.split("p")[1].split("a")[0] if "p" in v else "0"
        os.environ["SED_BF_ABL"] = v.split("a")[1] if "a" in v else "0"
        lib.sed_config_reload()
        for name, fn in ((f"c1 fused var{v}", c1_fused), (f"c2 fused var{v}", c2_fused)):
            res.setdefault(name, []).append(timeit(fn))
for k, v in res.items():
    v = sorted(v)
    print(f"{k:22s} median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}")
