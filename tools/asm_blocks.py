#!/usr/bin/env python3
"""Basic-block instruction histogram of one kernel in a hipcc -S listing.
    hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -o k.s file.hip ; python tools/asm_blocks.py k.s 'conv_pc_kernelILi64ELi32ELi2ELi1'"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z.*:", l) and pat in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blocks, cur = [], {"label": "entry", "ins": []}
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = {"label": m.group(1), "ins": []}
        continue
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        continue
    cur["ins"].append(s.split(";")[0].strip())
blocks.append(cur)


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_barrier", "s_nop", "s_sleep", "s_setprio")):
        return "misc"
    if op.startswith(("s_cbranch", "s_branch")):
        return "br"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


keys = ["mfma", "valu", "salu", "lds", "vmem", "smem", "wait", "misc", "br"]
print(f"{'block':>12s} {'n':>5s} " + " ".join(f"{k:>5s}" for k in keys) + "  branches")
tot = dict.fromkeys(keys, 0)
for b in blocks:
    c = dict.fromkeys(keys + ["other"], 0)
    tg = []
    for ins in b["ins"]:
        op = ins.split()[0]
        c[cls(op)] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            tg.append(ins.split()[-1])
    for k in keys:
        tot[k] += c[k]
    if len(b["ins"]) >= minn:
        print(f"{b['label']:>12s} {len(b['ins']):5d} " + " ".join(f"{c[k]:5d}" for k in keys) + "  " + ",".join(tg))
print(f"{'total':>12s} {sum(len(b['ins']) for b in blocks):5d} " + " ".join(f"{tot[k]:5d}" for k in keys))
