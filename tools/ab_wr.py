#!/usr/bin/env python3
"""Producer/consumer conv kernel: weights staged through the LDS (SED_PC_WR=0) against weights streamed from L2 into the consumers'
registers (default), interleaved in one process on the 128-channel shapes of the bench workload; also checks that both give the
same bits.   usage: ab_wr.py [rounds]"""
import os
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


def setup(H, W, Cin, Cout):
    d = dict(H=H, W=W, Cin=Cin, Cout=Cout)
    d["x"] = torch.randn(B, H, W, Cin, device=dev).to(bf)
    d["ref"] = torch.randn(B, H, W, Cout, device=dev).to(bf)
    d["out"] = torch.empty(B, H, W, Cout, device=dev, dtype=bf)
    d["sc_i"], d["sh_i"] = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    d["sc_o"], d["sh_o"] = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
    d["mean"], d["invstd"] = torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) + 0.5
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    d["wpack"] = torch.empty(9 * Cin * Cout, device=dev, dtype=bf)
    L.check(lib.sed_pack_conv_weight(1, P(w), P(d["wpack"]), Cout, Cin, Cout, Cin, 0, st))
    d["part"] = torch.zeros(lib.sed_conv_nparts(B, H, W) * 2 * max(Cin, Cout), device=dev)
    return d


def fwd_stats(d):      # forward, BN+ReLU prologue, statistics epilogue
    L.check(lib.sed_conv3x3_fwd(1, 1, 1, P(d["x"]), P(d["sc_i"]), P(d["sh_i"]), P(d["wpack"]), P(d["out"]), None, None, None, None, None,
                                P(d["part"]), B, d["H"], d["W"], d["Cin"], d["Cout"], st))


def dgrad_relu(d):     # data gradient form: no prologue, ReLU gate + BN-backward sums against the reference tile
    L.check(lib.sed_conv3x3_fwd(1, 0, 2, P(d["x"]), None, None, P(d["wpack"]), P(d["out"]), P(d["ref"]), P(d["sc_o"]), P(d["sh_o"]), P(d["mean"]),
                                P(d["invstd"]), P(d["part"]), B, d["H"], d["W"], d["Cin"], d["Cout"], st))


def store(d):
    L.check(lib.sed_conv3x3_fwd(1, 0, 0, P(d["x"]), None, None, P(d["wpack"]), P(d["out"]), None, None, None, None, None, None, B, d["H"], d["W"],
                                d["Cin"], d["Cout"], st))


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


def mode(v):
    os.environ["SED_PC_WR"] = v
    lib.sed_config_reload()


shapes = [(1500, 16, 64, 128), (1500, 16, 128, 128), (1500, 16, 128, 64), (750, 8, 128, 128), (3000, 32, 64, 128)]
for shp in shapes:
    d = setup(*shp)
    for name, fn in (("fwd bnrelu+stats", fwd_stats), ("dgrad relubwd", dgrad_relu), ("store", store)):
        outs = {}
        for v in ("0", "1"):
            mode(v)
            d["part"].zero_()
            fn(d)
            torch.cuda.synchronize()
            outs[v] = (d["out"].clone(), d["part"].clone())
        same = torch.equal(outs["0"][0], outs["1"][0])
        C2 = 2 * max(d["Cin"], d["Cout"]) if name != "store" else 1          # (the two forms cut the strips differently: compare column sums)
        n = (outs["0"][1].numel() // C2) * C2
        c0, c1 = outs["0"][1][:n].view(-1, C2).double().sum(0), outs["1"][1][:n].view(-1, C2).double().sum(0)
        pd = ((c0 - c1).abs().max() / max(1e-30, c0.abs().max().item())).item()
        t = {"0": [], "1": []}
        for r in range(rounds):
            for v in ("0", "1"):
                mode(v)
                t[v].append(timeit(fn, d))
        m0, m1 = sorted(t["0"])[rounds // 2], sorted(t["1"])[rounds // 2]
        gf = 2.0 * 9 * shp[2] * shp[3] * B * shp[0] * shp[1] / 1e9
        print(f"{shp[0]}x{shp[1]} {shp[2]}->{shp[3]} {name:18s} LDS weights {m0:.4f} ms   L2->registers {m1:.4f} ms ({gf / m1:.0f} TF/s)   "
              f"x{m0 / m1:.2f}   outputs equal: {same}   partial sums rel diff {pd:.1e}")
