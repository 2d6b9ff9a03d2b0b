#!/usr/bin/env python3
"""block-0 fused backward: consumer variants (SED_BC_TS = 0 | 1) interleaved in one process.  usage: ab_bc_ts.py [rounds]"""
import os
import runpy
import sys

sys.argv = [sys.argv[0], "0"] + sys.argv[2:] if len(sys.argv) > 1 else [sys.argv[0], "0"]
rounds = int(os.environ.get("AB_ROUNDS", "7"))
g = runpy.run_path(os.path.join(os.path.dirname(__file__), "bc_stamp.py"))
lib, timeit, fused, unfused = g["lib"], g["timeit"], g["fused"], g["unfused"]
res = {}
for r in range(rounds):
    for v in ("0", "1"):
        os.environ["SED_BC_TS"] = v
        lib.sed_config_reload()
        res.setdefault("fused TS=" + v, []).append(timeit(fused))
    res.setdefault("unfused", []).append(timeit(unfused))
for k, v in res.items():
    v = sorted(v)
    print(f"{k:20s} median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}")
