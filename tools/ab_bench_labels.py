#!/usr/bin/env python3
"""Runs bench.py (no CPU baseline) and prints ms/step plus the per-launch times of the labels matching a regular expression, and their sum:
the per-build command of tools/ab_build.sh / ab_multi.sh for changes that touch many launches.  usage: ab_bench_labels.py '<regex>' [steps]"""
import json
import re
import subprocess
import sys

rx = re.compile(sys.argv[1])
steps = sys.argv[2] if len(sys.argv) > 2 else "40"
out = subprocess.run([sys.executable, "bench.py", "--steps", steps, "--warmup", "8", "--no-cpu-baseline"], capture_output=True, text=True)
d = json.loads(out.stdout.strip().splitlines()[-1])
tot = 0.0
rows = []
for k, v in d["kernel_breakdown_ms"].items():
    if rx.search(k):
        ms = v["ms_total"] / d["steps"]
        tot += ms
        rows.append((ms, k))
print(f"step {d['ms_per_step']:.4f} ms   sum of kernels {d['gpu_time_ms_per_step_sum_of_kernels']:.4f}   matching labels {tot:.4f} ms")
for ms, k in sorted(rows, reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 0]:
    print(f"   {ms:.4f}  {k}")
