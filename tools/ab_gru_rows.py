#!/usr/bin/env python3
"""bf16 recurrence (biGRU-256, 16x16x32 form): rows per workgroup SED_GRU16_ROWS = 8 | 4 | 2 interleaved in one process on the CRNN's
shapes (t = 750); outputs compared bit for bit against the 4-row form.  usage: ab_gru_rows.py [B] [rounds]"""
import os
import sys

import torch

sys.path.insert(0, ".")
import sed_amd  # noqa: E402

L = sed_amd._lib
lib = L.lib()
P = L.ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t, Hd = 750, 256
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(5)
whh = [torch.randn(3 * Hd, Hd, device=dev, generator=g) * 0.06 for _ in range(2)]
pack_f = torch.empty(2 * 3 * Hd * Hd, device=dev, dtype=torch.bfloat16)
pack_b = torch.empty_like(pack_f)
L.check(lib.sed_gru_pack_weights(1, P(whh[0]), P(whh[1]), P(pack_f), P(pack_b), Hd, st))
gi = torch.randn(B, t, 6 * Hd, device=dev, generator=g)
bhh = torch.randn(2, 3 * Hd, device=dev, generator=g) * 0.1
dh = torch.randn(B, t, 2 * Hd, device=dev, generator=g) * 0.1
hseq = torch.empty(B, t, 2 * Hd, device=dev)
saved = torch.empty(B, t, 8 * Hd, device=dev)
dgi = torch.empty(B, t, 6 * Hd, device=dev)
dgh = torch.empty(B, t, 6 * Hd, device=dev)


def fwd():
    L.check(lib.sed_gru_seq_fwd(1, P(gi), P(bhh), P(pack_f), P(hseq), P(saved), B, t, Hd, st))


def bwd():
    L.check(lib.sed_gru_seq_bwd(1, P(dh), P(hseq), P(saved), P(pack_b), P(dgi), P(dgh), B, t, Hd, st))


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


outs, res = {}, {}
for r in range(rounds):
    for v in ("8", "4", "2"):
        os.environ["SED_GRU16_ROWS"] = v
        lib.sed_config_reload()
        res.setdefault(v, [[], []])
        res[v][0].append(timeit(fwd))
        res[v][1].append(timeit(bwd))
        if r == 0:
            torch.cuda.synchronize()
            outs[v] = [x.clone() for x in (hseq, saved, dgi, dgh)]
for v, (a, b) in res.items():
    a, b = sorted(a), sorted(b)
    same = all(torch.equal(x, y) for x, y in zip(outs[v], outs["4"]))
    md = max(float((x - y).abs().max() / y.abs().max()) for x, y in zip(outs[v], outs["4"]))
    print(f"B={B} SED_GRU16_ROWS={v}: forward {a[len(a) // 2]:.4f} ms   backward {b[len(b) // 2]:.4f} ms   bit-identical to the 4-row form: {same}"
          f"   max |diff| / max |ref| {md:.1e}")
