#!/usr/bin/env python3
"""Upper bound for "the data-gradient kernel produces dz" (round 4): the weight gradient of blocks 2-3 with dz GIVEN (plain loads) against
the fused form that produces dz = BatchNorm / ReLU / pool backward on load and writes it out, interleaved on one device."""
import runpy
import sys

sys.argv = [sys.argv[0], "0"]
g = runpy.run_path("experiments/tools/ab_fused_cs.py")
L, lib, P, torch, B, st, timeit = g["L"], g["lib"], g["P"], g["torch"], g["B"], g["st"], g["timeit"]
layers = g["layers"]


def w_fused_c1(d):
    H, W, Ci, Co = d["H"], d["W"], d["Cin"], d["Cout"]
    L.check(lib.sed_conv3x3_wgrad_fused_u(1, 0, P(d["x"]), None, None, 2, P(d["g"]), P(d["z"]), None, None, P(d["ca"]), P(d["cb"]), P(d["cc"]), 1,
                                          P(d["dz"]), P(d["dwp"]), P(d["ws"]), B, H, W, Ci, Co, P(d["dw"]), Co, Ci, st))


def w_fused_c2(d):
    H, W, Ci, Co, pool = d["H"], d["W"], d["Cin"], d["Cout"], d["pool"]
    L.check(lib.sed_conv3x3_wgrad_fused_u(1, 1, P(d["x"]), P(d["sc_i"]), P(d["sh_i"]), 1, P(d["dy"]), P(d["z"]), P(d["sc_o"]), P(d["sh_o"]), P(d["ca"]),
                                          P(d["cb"]), P(d["cc"]), pool, P(d["dz"]), P(d["dwp"]), P(d["ws"]), B, H, W, Ci, Co, P(d["dw"]), Co, Ci, st))


def w_plain(pro):
    def f(d):
        H, W, Ci, Co = d["H"], d["W"], d["Cin"], d["Cout"]
        L.check(lib.sed_conv3x3_wgrad(1, pro, P(d["x"]), P(d["sc_i"]) if pro else None, P(d["sh_i"]) if pro else None, P(d["dz"]), P(d["dwp"]),
                                      P(d["ws"]), B, H, W, Ci, Co, st))
    return f


res = {}
for r in range(5):
    for i, (name, d, _, _) in enumerate(layers):
        c2 = i % 2 == 1
        res.setdefault(name + " wgrad fused (produces + writes dz)", []).append(timeit(w_fused_c2 if c2 else w_fused_c1, d))
        res.setdefault(name + " wgrad plain (dz given)", []).append(timeit(w_plain(1 if c2 else 0), d))
for k, v in res.items():
    v = sorted(v)
    print(f"{k:66s} median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}")
