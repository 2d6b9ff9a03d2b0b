#!/usr/bin/env python3
"""Block 0 (C1 mode) at the bench geometry, interleaved in one process: conv1's ReLU decisions as a bit mask written by the forward and
read by the fused backward (round 4) against the gate DERIVED in the backward from its own rebuilt activation tile (round 5:
relu_mask = NULL in both calls).  Forward mask -> backward consistency: the derived run must reproduce the masked run bit for bit.
usage: ab_c1_gate.py [rounds]"""
import sys

import torch

sys.path.insert(0, ".")
import sed_amd  # noqa: E402

L = sed_amd._lib
lib = L.lib()
P = L.ptr
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev, bf = "cuda", torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
B, H, W, C = 32, 6001, 64, 32
g = torch.Generator(device=dev).manual_seed(5)
x1 = torch.randn(B, H, W, device=dev, generator=g) * 3 + 1
fmean, fstd = torch.randn(W, device=dev, generator=g), torch.rand(W, device=dev, generator=g) + 0.5
w1 = torch.randn(C, 1, 3, 3, device=dev, generator=g) * 0.4
sc1, sh1 = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.3
dy = torch.randn(B, H // 2, W // 2, C, device=dev, generator=g).to(bf)
sc2, sh2 = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.3
ca, cb, cc = torch.randn(C, device=dev, generator=g), torch.randn(C, device=dev, generator=g) * 0.1, torch.randn(C, device=dev, generator=g) * 0.1
w2 = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05
wp = torch.empty(9 * C * C, device=dev, dtype=bf)
wt = torch.empty(9 * C * C, device=dev, dtype=bf)
L.check(lib.sed_pack_conv_weight(1, P(w2), P(wp), C, C, C, C, 0, st))
L.check(lib.sed_pack_conv_weight(1, P(w2), P(wt), C, C, C, C, 1, st))
mask = torch.zeros(B, H, W, 2, device=dev, dtype=torch.int16)
z2 = torch.empty(B, H, W, C, device=dev, dtype=bf)
fpart = torch.empty(lib.sed_conv_nparts(B, H, W), 2, C, device=dev)
npart = lib.sed_conv_dgrad_c1_nparts()
part = torch.empty(npart, 10, C, device=dev)
ws = torch.empty(lib.sed_conv_wgrad_ws_floats(B, H, W, C, C), device=dev)
dwp = torch.empty(9 * C * C, device=dev)
dw = torch.empty(C, C, 3, 3, device=dev)


def fwd(m):
    L.check(lib.sed_conv3x3_fwd_c1(1, 1, P(x1), P(fmean), P(fstd), P(w1), P(sc1), P(sh1), P(wp), P(z2), P(fpart), P(mask) if m else None,
                                   B, H, W, C, st))


def bwd(m):
    L.check(lib.sed_conv3x3_bwd_fused_c1(1, P(x1), P(fmean), P(fstd), P(w1), P(sc1), P(sh1), P(dy), P(z2), P(sc2), P(sh2), P(ca), P(cb), P(cc), 2,
                                         P(wt), P(mask) if m else None, P(part), P(dwp), P(ws), B, H, W, C, P(dw), C, C, st))


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


fwd(True)
bwd(True)
torch.cuda.synchronize()
ref = (part.clone(), dw.clone(), z2.clone())
fwd(False)
bwd(False)
torch.cuda.synchronize()
print("derived gate == mask given (bit for bit): [A; sum g]", torch.equal(part, ref[0]), " dW2", torch.equal(dw, ref[1]), " z2", torch.equal(z2, ref[2]))

res = {}
for r in range(rounds):
    for name, fn in (("forward, mask written", lambda: fwd(True)), ("forward, no mask", lambda: fwd(False)),
                     ("fused backward, mask given (+ wgrad reduce)", lambda: bwd(True)), ("fused backward, gate derived (+ wgrad reduce)", lambda: bwd(False))):
        res.setdefault(name, []).append(timeit(fn))
for k, v in res.items():
    v = sorted(v)
    print(f"{k:50s} median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}")
