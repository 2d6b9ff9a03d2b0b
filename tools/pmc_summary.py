#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (name + grid), per-wave
instruction mix and wait fractions.  usage: pmc_summary.py <csv> [<csv> ...] [--filter substr]"""
import collections
import csv
import sys

args = sys.argv[1:]
flt = None
if "--filter" in args:
    i = args.index("--filter")
    flt = args[i + 1]
    del args[i:i + 2]
files = args
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt and flt not in k:
            continue
        key = (k.split("(")[0][-60:], r["Grid_Size"], r["Workgroup_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(key, r["Counter_Name"])].add(r["Dispatch_Id"])
for key, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = {cn: max(1, len(disp[(key, cn)])) for cn in c}
    g = lambda cn: c.get(cn, 0.0) / n.get(cn, 1)
    waves = g("SQ_WAVES") or 1
    wc = g("SQ_WAVE_CYCLES") or 1
    print(f"\n{key[0]}  grid={key[1]} wg={key[2]}  dispatches={n.get('SQ_WAVE_CYCLES', 0)}")
    print("  waves %.0f  wave_cycles/wave %.0f  wait_any %.0f%%  wait_inst %.0f%%  act_valu %.0f%%  act_lds %.0f%%  act_vmem %.0f%%" % (
        waves, wc / waves * 4, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc,
        100 * g("SQ_ACTIVE_INST_VALU") / wc, 100 * g("SQ_ACTIVE_INST_LDS") / wc, 100 * g("SQ_ACTIVE_INST_VMEM") / wc))
    if "SQ_INSTS_VALU" in c:
        vm = (g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")) or 1
        print("  per wave: valu %.0f  mfma %.0f  lds %.0f  vmem_rd %.0f  vmem_wr %.0f  salu %.0f | avg vmem latency %.0f cyc  avg lds latency %.0f cyc" % (
            g("SQ_INSTS_VALU") / waves, g("SQ_INSTS_MFMA") / waves, g("SQ_INSTS_LDS") / waves, g("SQ_INSTS_VMEM_RD") / waves,
            g("SQ_INSTS_VMEM_WR") / waves, g("SQ_INSTS_SALU") / waves, g("SQ_INST_LEVEL_VMEM") / vm, g("SQ_INST_LEVEL_LDS") / (g("SQ_INSTS_LDS") or 1)))
