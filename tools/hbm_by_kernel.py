#!/usr/bin/env python3
"""HBM bytes per launch BY KERNEL NAME from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, FETCH x2 on gfx950 as
MI355X_MICROARCH.md prescribes) joined with the average launch duration of a kernel-trace run of the same command.
usage: hbm_by_kernel.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel_trace.csv> <out.json> > table.txt"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    tot, ids = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            ids[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (tot[k] / max(1, len(ids[k])), len(ids[k])) for k in tot}


def short(name):
    return re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", name))[:110]


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[3])):
    dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out, rows = {}, []
for k in fetch:
    rd = fetch[k][0] * 1024.0 * 2.0
    wr = write.get(k, (0.0, 0))[0] * 1024.0
    d = dur.get(k)
    us = sum(d) / len(d) / 1e3 if d else None
    out[k] = {"kernel": short(k), "hbm_bytes_per_launch": rd + wr, "read": rd, "write": wr, "avg_us": us, "launches_profiled": fetch[k][1]}
    rows.append((us * len(d) if d else 0.0, us or 0.0, k, rd, wr, len(d) if d else 0))
json.dump(out, open(sys.argv[4], "w"), indent=1)
tot_ns = sum(r[0] for r in rows)
print(f"{'avg us':>9} {'launches':>8} {'share':>6} {'read GB':>8} {'write GB':>8} {'TB/s':>6} {'of 8':>5}  kernel")
for tot, us, k, rd, wr, n in sorted(rows, reverse=True)[:40]:
    tb = (rd + wr) / (us * 1e-6) / 1e12 if us else 0.0
    print(f"{us:9.1f} {n:8d} {tot / max(tot_ns, 1e-9):6.3f} {rd / 1e9:8.4f} {wr / 1e9:8.4f} {tb:6.2f} {tb / 8:5.2f}  {short(k)}")
