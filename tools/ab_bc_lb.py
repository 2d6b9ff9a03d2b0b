#!/usr/bin/env python3
"""block-0 fused backward: who rebuilds the conv1 tile (SED_BC_LB = 0 consumers | 1 loader waves), interleaved in one process; the
outputs of the two variants are compared bit for bit.  usage: ab_bc_lb.py   (AB_ROUNDS=7)"""
import os
import runpy
import sys

sys.argv = [sys.argv[0], "0"]
rounds = int(os.environ.get("AB_ROUNDS", "7"))
g = runpy.run_path(os.path.join(os.path.dirname(__file__), "bc_stamp.py"))
lib, timeit, fused, torch = g["lib"], g["timeit"], g["fused"], g["torch"]
outs = {}
res = {}
for r in range(rounds):
    for v in ("0", "1"):
        os.environ["SED_BC_LB"] = v
        lib.sed_config_reload()
        res.setdefault("fused LB=" + v, []).append(timeit(fused))
        if r == 0:
            torch.cuda.synchronize()
            outs[v] = (g["dw"].clone(), g["part"].clone())
for k, v in res.items():
    v = sorted(v)
    print(f"{k:20s} median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}")
print("dW identical:", bool(torch.equal(outs["0"][0], outs["1"][0])), " [A; sum g] partials identical:", bool(torch.equal(outs["0"][1], outs["1"][1])))
