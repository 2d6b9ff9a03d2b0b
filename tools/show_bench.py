#!/usr/bin/env python3
"""Pretty-print a bench.py JSON line: headline + per-kernel ms/step."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{d['value']:.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  n_gpus={d['n_gpus']}  dtype={d['dtype']}")
print("roofline:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["roofline"].items()})
print("cpu_baseline:", d.get("cpu_baseline"))
tot = 0.0
for k, v in d["kernel_breakdown_ms"].items():
    ms = v["ms_total"] / d["steps"]
    tot += ms
    print(f"{ms:8.3f} ms  x{v['n'] // d['steps']:<2d} {k}")
print(f"{tot:8.3f} ms  sum of kernels")
