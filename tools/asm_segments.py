#!/usr/bin/env python3
"""Instruction mix of a kernel's .s between consecutive s_barrier instructions (what a stage of each wave role carries).
usage: asm_segments.py file.s [kernel-name-substring]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):\s.*?^\.Lfunc_end\d+:", txt, re.S | re.M):
    name = m.group(1)
    if sub not in name:
        continue
    lines = [l.strip() for l in m.group(0).split("\n")[1:]]
    lines = [l for l in lines if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
    segs, cur = [], []
    for l in lines:
        cur.append(l)
        if l.startswith("s_barrier"):
            segs.append(cur)
            cur = []
    segs.append(cur)
    print(name[-60:], "instructions", len(lines), "segments", len(segs))
    for k, sg in enumerate(segs):
        c = collections.Counter()
        for l in sg:
            op = l.split()[0]
            key = ("mfma" if op.startswith("v_mfma") else "ds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "scratch_"))
                   else "valu" if op.startswith("v_") else "wait" if op.startswith("s_waitcnt") else "branch" if op.startswith(("s_cbranch", "s_branch"))
                   else "salu" if op.startswith("s_") else "other")
            c[key] += 1
        print(f"  seg {k:2d} {len(sg):5d}", dict(c))
