#!/usr/bin/env python3
"""What sits between consecutive MFMAs of a kernel's main loop block (hipcc -S listing).
    python tools/asm_gaps.py k.s KERNEL_SUBSTRING [BLOCK_LABEL]      prints per gap: V=valu S=salu L=lds-read W=lds-write M=vmem w=s_waitcnt n=s_nop"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
st = next(i for i, l in enumerate(lines) if re.match(r"^_Z.*:", l) and pat in l)
en = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
blocks, cur, name = {}, [], "entry"
for l in lines[st + 1:en]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks[name] = cur
        name, cur = m.group(1), []
        continue
    s = l.split(";")[0].strip()
    if s and not s.startswith("."):
        cur.append(s)
blocks[name] = cur
label = sys.argv[3] if len(sys.argv) > 3 else max(blocks, key=lambda k: sum(1 for x in blocks[k] if x.startswith("v_mfma")))
gap, out = "", []
for ins in blocks[label]:
    op = ins.split()[0]
    if op.startswith("v_mfma"):
        out.append(gap)
        gap = ""
    elif op.startswith("ds_read") or op.startswith("ds_load"):
        gap += "L"
    elif op.startswith("ds_"):
        gap += "W"
    elif op.startswith(("buffer_", "global_", "scratch_")):
        gap += "M"
    elif op.startswith("s_waitcnt"):
        gap += "w"
    elif op.startswith("s_nop"):
        gap += "n"
    elif op.startswith("s_barrier"):
        gap += "B"
    elif op.startswith("s_"):
        gap += "S"
    elif op.startswith("v_"):
        gap += "V"
    else:
        gap += "?"
out.append(gap)
print(label, len(out) - 1, "MFMAs")
for i, g in enumerate(out):
    print(f"{i:3d} {len(g):3d} {g}")
