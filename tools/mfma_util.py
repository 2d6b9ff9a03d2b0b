#!/usr/bin/env python3
"""Per-kernel MFMA utilisation and effective clock from tools/pmc_mfma.sh:
   util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * kernel cycles),  kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs,
   clock = kernel cycles / kernel duration (kernel trace of a second run; MI355X_MICROARCH.md 'DVFS give-back').
usage: mfma_util.py <counter_collection.csv> <kernel_trace.csv>"""
import collections
import csv
import sys

cnt = collections.defaultdict(lambda: collections.defaultdict(float))
nd = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
    nd[k].add(r["Dispatch_Id"])
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
rows = []
for k, c in cnt.items():
    n = len(nd[k])
    if k not in dur or not c.get("GRBM_GUI_ACTIVE"):
        continue
    d_ns = sum(dur[k]) / len(dur[k])
    cyc = c["GRBM_GUI_ACTIVE"] / n / 8.0
    util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n / (1024.0 * cyc)
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    rows.append((d_ns, k, cyc / d_ns, util, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_VALU", 0) / wc))
print(f"{'avg us':>8} {'GHz':>5} {'MFMA util':>9} {'wait_any':>8} {'wait_inst':>9} {'valu act':>8}  kernel")
for d_ns, k, ghz, util, wa, wi, va in sorted(rows, reverse=True)[:40]:
    print(f"{d_ns / 1e3:8.1f} {ghz:5.2f} {100 * util:8.1f}% {100 * wa:7.1f}% {100 * wi:8.1f}% {100 * va:7.1f}%  {k[:110]}")
