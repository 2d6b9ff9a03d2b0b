#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 counter_collection.csv, optionally as ratios to one of them.
    python tools/pmc_table.py FILE.csv [DENOMINATOR_COUNTER]"""
import collections
import csv
import sys

path = sys.argv[1]
den = sys.argv[2] if len(sys.argv) > 2 else None
disp = {}
for r in csv.DictReader(open(path)):
    d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in disp.values():
    a = agg[d["name"]]
    a["n"] += 1
    for k, v in d.items():
        if k != "name":
            a[k] += v
names = sorted({k for a in agg.values() for k in a if k not in ("n", "ns")})
rows = sorted(agg.items(), key=lambda kv: -kv[1]["ns"])
print(f"{'us':>8s} " + " ".join(f"{n[-14:]:>14s}" for n in names) + "  kernel")
for name, a in rows:
    us = a["ns"] / a["n"] / 1e3
    if us < 20:
        continue
    vals = []
    for n in names:
        v = a[n] / a["n"]
        if den and n != den:
            vals.append(f"{v / max(a[den] / a['n'], 1e-9):14.3f}")
        else:
            vals.append(f"{v:14.4g}")
    print(f"{us:8.1f} " + " ".join(vals) + f"  {name[:90]}")
