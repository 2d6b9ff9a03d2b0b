#!/usr/bin/env python3
"""sha256 over the CODE of the kernel sources: csrc/*.hip, csrc/*.h, include/sed_hip.h with comments removed and white space
collapsed (string and character literals kept as they are), csrc/Makefile without its `#` comments -- sorted by name, name + text.
tools/hbm_traffic.py stamps profiles/hbm_traffic_by_label.json with it at profiling time; bench.py recomputes it and reports
`traffic: null, traffic_stale: true` when the tree's kernels are no longer the ones the PMC run measured.  Round 5: the hash no
longer covers comments, so documentation of the ABI header or of a kernel can change without invalidating a measurement
("weak" 12: a caveat had been removed from include/sed_hip.h to keep the stamp)."""
import glob
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_TOKEN = re.compile(r'''
      //[^\n]*                      # line comment
    | /\*.*?\*/                     # block comment
    | "(?:\\.|[^"\\\n])*"           # string literal
    | '(?:\\.|[^'\\\n])*'           # character literal
''', re.S | re.X)


def strip_c(text):
    """C / C++ / HIP source without comments, runs of white space collapsed to one blank, blank lines dropped."""
    def repl(m):
        t = m.group(0)
        return " " if t.startswith("/") else t          # a comment separates tokens like a blank
    text = text.replace("\\\n", " ")                    # line continuations (macros; also inside // comments)
    text = _TOKEN.sub(repl, text)
    lines = [re.sub(r"[ \t\r\f\v]+", " ", ln).strip() for ln in text.split("\n")]
    return "\n".join(ln for ln in lines if ln)


def strip_make(text):
    lines = [re.sub(r"[ \t]+", " ", re.sub(r"(^|\s)#.*$", "", ln)).rstrip() for ln in text.split("\n")]
    return "\n".join(ln for ln in lines if ln.strip())


def csrc_sha256(root=ROOT):
    cs = os.path.join(root, "soundeventdetection-pytorch_amd", "csrc")
    files = sorted(glob.glob(os.path.join(cs, "*.hip")) + glob.glob(os.path.join(cs, "*.h")) + [os.path.join(cs, "Makefile"),
                   os.path.join(root, "include", "sed_hip.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "r", encoding="utf-8", errors="replace") as fh:
            text = fh.read()
        h.update((strip_make(text) if f.endswith("Makefile") else strip_c(text)).encode())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256())
