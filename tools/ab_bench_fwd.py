#!/usr/bin/env python3
"""one short bench.py run, prints the step time and the per-launch times of the forward convolutions (A/B builds: tools/ab_build.sh FLAG "python tools/ab_bench_fwd.py")"""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith('{"metric"')][0])
kb = d["kernel_breakdown_ms"]
pat = sys.argv[1] if len(sys.argv) > 1 else ":fwd"
rows = [(k, v["ms_total"] / v["n"]) for k, v in kb.items() if pat in k and k.startswith("sed_conv3x3")]
rows.sort(key=lambda kv: kv[0])
tot = sum(ms for _, ms in rows)
print(f"step {d['ms_per_step']:.3f} ms   sum of '{pat}' conv launches {tot:.3f} ms   " + "  ".join(f"{k.split(':')[1].split(' H')[0]}={ms:.3f}" for k, ms in rows))
