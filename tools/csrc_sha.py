#!/usr/bin/env python3
"""sha256 over the kernel sources (csrc/*.hip, csrc/*.h, csrc/Makefile, include/sed_hip.h; sorted by name, name + bytes).
tools/hbm_traffic.py stamps profiles/hbm_traffic_by_label.json with it at profiling time; bench.py recomputes it and reports
`traffic: null, traffic_stale: true` when the tree's kernels are no longer the ones the PMC run measured."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256(root=ROOT):
    cs = os.path.join(root, "soundeventdetection-pytorch_amd", "csrc")
    files = sorted(glob.glob(os.path.join(cs, "*.hip")) + glob.glob(os.path.join(cs, "*.h")) + [os.path.join(cs, "Makefile"),
                   os.path.join(root, "include", "sed_hip.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256())
