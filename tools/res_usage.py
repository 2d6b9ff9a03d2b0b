#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks (stderr of a compile): one line per kernel.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> res.txt; tools/res_usage.py res.txt [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = []
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(.*?): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.splitlines() if rows else []
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n).split("(")[0]
    if flt and flt not in n:
        continue
    print(f"{n:60s} vgpr {r.get('VGPRs', -1):4d} agpr {r.get('AGPRs', -1):4d} spill {r.get('VGPRs Spill', -1):4d} scratch {r.get('ScratchSize [bytes/lane]', -1):4d} sgpr {r.get('TotalSGPRs', -1):4d}")
