#!/usr/bin/env python3
"""block-0 fused backward: issue priority of the loader waves (SED_BC_PRIO = 0..3), interleaved in one process."""
import os
import runpy
import sys

sys.argv = [sys.argv[0], "0"]
rounds = int(os.environ.get("AB_ROUNDS", "5"))
g = runpy.run_path(os.path.join(os.path.dirname(__file__), "bc_stamp.py"))
lib, timeit, fused = g["lib"], g["timeit"], g["fused"]
res = {}
for r in range(rounds):
    for v in ("0", "1", "2", "3"):
        os.environ["SED_BC_PRIO"] = v
        lib.sed_config_reload()
        res.setdefault("loader priority " + v, []).append(timeit(fused))
for k, v in res.items():
    v = sorted(v)
    print(f"{k:20s} median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}")
