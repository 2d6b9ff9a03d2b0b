#!/usr/bin/env python3
"""Blocks 2-3 backward at the bench geometry (B = 32): "the weight gradient produces dz" (round 3: sed_conv3x3_wgrad_fused_u writes dz,
the data gradient reads it) against "the data gradient produces dz" (round 4: sed_conv3x3_dgrad_dz writes dz, sed_conv3x3_wgrad_u reads
it), interleaved on ONE device; every output of the two orders is compared bit for bit.   usage: ab_dgrad_dz.py [rounds]"""
import runpy
import sys

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sys.argv = [sys.argv[0], "0"]
g = runpy.run_path("experiments/tools/ab_fused_cs.py")
L, lib, P, torch, B, st, timeit = g["L"], g["lib"], g["P"], g["torch"], g["B"], g["st"], g["timeit"]
layers = g["layers"]


def new_c1(d):
    H, W, Ci, Co = d["H"], d["W"], d["Cin"], d["Cout"]
    L.check(lib.sed_conv3x3_dgrad_dz(1, 2, P(d["g"]), P(d["z"]), None, None, P(d["ca"]), P(d["cb"]), P(d["cc"]), 1, P(d["wt"]), P(d["dz"]), P(d["dx"]), 4,
                                     P(d["x"]), P(d["cnt"]), P(d["sc_i"]), P(d["sh_i"]), P(d["mu_i"]), P(d["is_i"]), P(d["part"]), d["np"], P(d["flag"]),
                                     B, H, W, Co, Ci, st))
    L.check(lib.sed_conv3x3_wgrad_u(1, 0, P(d["x"]), None, None, P(d["dz"]), P(d["dwp"]), P(d["ws"]), B, H, W, Ci, Co, P(d["dw"]), Co, Ci, st))


def new_c2(d):
    H, W, Ci, Co, pool = d["H"], d["W"], d["Cin"], d["Cout"], d["pool"]
    L.check(lib.sed_conv3x3_dgrad_dz(1, 1, P(d["dy"]), P(d["z"]), P(d["sc_o"]), P(d["sh_o"]), P(d["ca"]), P(d["cb"]), P(d["cc"]), pool, P(d["wt"]), P(d["dz"]),
                                     P(d["dx"]), 2, P(d["x"]), None, P(d["sc_i"]), P(d["sh_i"]), P(d["mu_i"]), P(d["is_i"]), P(d["part"]), d["np"], None,
                                     B, H, W, Co, Ci, st))
    L.check(lib.sed_conv3x3_wgrad_u(1, 1, P(d["x"]), P(d["sc_i"]), P(d["sh_i"]), P(d["dz"]), P(d["dwp"]), P(d["ws"]), B, H, W, Ci, Co, P(d["dw"]), Co, Ci, st))


def snap(d):
    torch.cuda.synchronize()
    return [d[k].clone() for k in ("dz", "dx", "dw", "dwp")] + [d["part"][: d["np"] * 2 * d["Cin"]].clone()]


res, same = {}, {}
for r in range(rounds):
    for i, (name, d, old, _) in enumerate(layers):
        new = new_c2 if i % 2 else new_c1
        res.setdefault(name + "  wgrad produces dz", []).append(timeit(old, d))
        if r == 0:
            a = snap(d)
            for k in ("dz", "dx", "dw", "dwp", "part"):
                d[k].fill_(7.0)
        res.setdefault(name + "  dgrad produces dz", []).append(timeit(new, d))
        if r == 0:
            b = snap(d)
            same[name] = [bool(torch.equal(x, y)) for x, y in zip(a, b)]
tot = [0.0, 0.0]
for k, v in res.items():
    v = sorted(v)
    print(f"{k:50s} median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}")
    tot[k.endswith("dgrad produces dz")] += v[len(v) // 2]
print("sum of medians: wgrad produces dz %.4f ms, dgrad produces dz %.4f ms" % tuple(tot))
for k, v in same.items():
    print(f"{k:32s} bit-identical (dz, dx, dW, dW packed, statistics partials): {v}")
