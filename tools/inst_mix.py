#!/usr/bin/env python3
"""Per-kernel instruction mix and issue rate from a rocprofv3 --pmc counter_collection.csv.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA \
              -d DIR -o NAME --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
    python tools/inst_mix.py DIR/.../NAME_counter_collection.csv [clock_GHz]

SQ_INSTS_* count wave-instructions summed over the chip; the duration is the dispatch's own Start/End timestamp in the
same CSV (dispatches are serialised under counter collection, so it is the kernel's solo time).  The issue rate is
wave-instructions / (duration x clock x 1024 SIMDs): 1.0 would be one instruction per SIMD per cycle."""
import collections
import csv
import sys

path = sys.argv[1]
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
SIMDS = 1024

disp = {}
for r in csv.DictReader(open(path)):
    d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                           "vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "lds": r["LDS_Block_Size"]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])

agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in disp.values():
    a = agg[d["name"]]
    a["n"] += 1
    for k, v in d.items():
        if k not in ("name", "vgpr", "agpr", "lds"):
            a[k] += v
    a["vgpr"], a["agpr"], a["lds"] = d["vgpr"], d["agpr"], d["lds"]

rows = []
for name, a in agg.items():
    n = a["n"]
    us = a["ns"] / n / 1e3
    g = lambda k: a.get(k, 0.0) / n / 1e6                                    # noqa: E731
    vmem = g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")
    tot = g("SQ_INSTS_VALU") + g("SQ_INSTS_SALU") + g("SQ_INSTS_LDS") + vmem   # MFMA is part of VALU on gfx950
    ipc = tot * 1e6 / (us * 1e3 * ghz * SIMDS) if us > 0 else 0.0
    rows.append((us * n, us, int(n), g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), vmem, g("SQ_INSTS_MFMA"), ipc,
                 a["vgpr"], a["agpr"], a["lds"], name))
rows.sort(reverse=True)
print(f"wave-instructions per launch (millions); issue rate = instr / (duration x {ghz} GHz x {SIMDS} SIMDs)")
print(f"{'us':>8s} {'calls':>5s} {'VALU':>7s} {'SALU':>7s} {'LDS':>7s} {'VMEM':>6s} {'MFMA':>6s} {'instr/cyc/SIMD':>14s} {'vgpr':>4s} {'agpr':>4s} {'lds':>6s}  kernel")
for tot_us, us, n, va, sa, ld, vm, mf, ipc, vg, ag, lds, name in rows:
    if us < 3.0:
        continue
    print(f"{us:8.1f} {n:5d} {va:7.2f} {sa:7.2f} {ld:7.2f} {vm:6.2f} {mf:6.2f} {ipc:14.3f} {vg:>4s} {ag:>4s} {lds:>6s}  {name[:96]}")
