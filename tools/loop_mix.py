#!/usr/bin/env python3
"""Static instruction mix of the big loops of every kernel in a hipcc -S listing: for each loop body longer than MIN lines the counts of
MFMA / VALU / SALU / LDS / VMEM / waitcnt / branch instructions (all paths of the body, so conditional blocks count once).
usage: hipcc ... -S --cuda-device-only x.hip -o x.s ; tools/loop_mix.py x.s [min_lines] [kernel filter]"""
import re
import subprocess
import sys

lines = open(sys.argv[1]).read().split("\n")
minlen = int(sys.argv[2]) if len(sys.argv) > 2 else 300
flt = sys.argv[3] if len(sys.argv) > 3 else ""


def cat(l):
    l = l.strip()
    if not l or l.startswith(".") or l.startswith(";"):
        return None
    op = l.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


kern = None
kstart = {}
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kern = m.group(1)
        kstart[i] = kern
names = {}
if kstart:
    out = subprocess.run(["c++filt"] + list(kstart.values()), capture_output=True, text=True).stdout.splitlines()
    names = dict(zip(kstart.values(), out))
# basic blocks: label line -> next label line; a block belongs to loop H when its label comment says "Loop Header" (H itself) or
# "in Loop: Header=BBx_y" (hipcc rotates loops: the back edge may target an inner block, so branch targets are not used)
lab_re = re.compile(r"^(\.LBB\d+_\d+):(.*)$")
blocks = []          # (start, end, label, comment, kernel)
cur = None
starts = []
for i, l in enumerate(lines):
    if i in kstart:
        cur = kstart[i]
    m = lab_re.match(l)
    if m:
        starts.append((i, m.group(1), m.group(2), cur))
for n, (i, lab, com, k) in enumerate(starts):
    end = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
    for q in range(i, end):
        if "s_endpgm" in lines[q]:
            end = q + 1
            break
    blocks.append((i, end, lab, com, k))
loops = {}
order = []
for (i, end, lab, com, k) in blocks:
    m = re.search(r"in Loop: Header=BB(\d+_\d+) Depth=1", com)
    h = None
    if "Loop Header: Depth=1" in com:
        h = lab
    elif m:
        h = ".LBB" + m.group(1)
    if h is None:
        continue
    key = (k, h)
    if key not in loops:
        loops[key] = []
        order.append(key)
    loops[key].append((i, end))
for key in order:
    k, h = key
    n = sum(e - i for i, e in loops[key])
    if n < minlen:
        continue
    nm = re.sub(r"\(anonymous namespace\)::", "", names.get(k, k or "?")).split("(")[0]
    if flt and flt not in nm:
        continue
    cnt = {}
    for (i, e) in loops[key]:
        for x in lines[i:e]:
            c = cat(x)
            if c:
                cnt[c] = cnt.get(c, 0) + 1
    tot = sum(cnt.values())
    print(f"{nm[:70]:70s} {h:12s} len {n:5d}  " + "  ".join(f"{c} {cnt.get(c, 0)}" for c in ("mfma", "valu", "salu", "lds", "vmem", "wait", "branch", "nop")) + f"   salu share {100 * cnt.get('salu', 0) / max(1, tot):.0f} %")
