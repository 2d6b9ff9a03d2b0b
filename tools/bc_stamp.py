#!/usr/bin/env python3
"""block-0 backward at the bench geometry: the two-kernel form against sed_conv3x3_bwd_fused_c1, interleaved in one process
(phase stamps print with a STAMPS=1 build).  usage: bc_stamp.py [rounds]"""
import sys

import torch

sys.path.insert(0, ".")
import sed_amd  # noqa: E402

L = sed_amd._lib
lib = L.lib()
P = L.ptr
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev, bf = "cuda", torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
B, H, W, C = 32, 6001, 64, 32
x1 = torch.randn(B, H, W, device=dev)
fmean, fstd = torch.randn(W, device=dev), torch.rand(W, device=dev) + 0.5
w1 = torch.randn(C, 1, 3, 3, device=dev) * 0.4
sc1, sh1 = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
z2 = torch.randn(B, H, W, C, device=dev).to(bf)
dy = torch.randn(B, H // 2, W // 2, C, device=dev).to(bf)
sc2, sh2 = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
ca, cb, cc = torch.randn(C, device=dev), torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
w2 = torch.randn(C, C, 3, 3, device=dev) * 0.05
wt = torch.empty(9 * C * C, device=dev, dtype=bf)
L.check(lib.sed_pack_conv_weight(1, P(w2), P(wt), C, C, C, C, 1, st))
mask = torch.randint(0, 65536, (B, H, W, 2), device=dev, dtype=torch.int32).to(torch.int16)
npart = lib.sed_conv_dgrad_c1_nparts()
part = torch.empty(npart, 10, C, device=dev)
ws = torch.empty(lib.sed_conv_wgrad_ws_floats(B, H, W, C, C), device=dev)
dwp = torch.empty(9 * C * C, device=dev)
dw = torch.empty(C, C, 3, 3, device=dev)
dz = torch.empty(B, H, W, C, device=dev, dtype=bf)


def unfused():
    L.check(lib.sed_conv3x3_wgrad_fused_c1_u(1, P(x1), P(fmean), P(fstd), P(w1), P(sc1), P(sh1), P(dy), P(z2), P(sc2), P(sh2), P(ca), P(cb), P(cc),
                                             2, P(dz), P(dwp), P(ws), B, H, W, C, P(dw), C, C, st))
    L.check(lib.sed_conv3x3_dgrad_c1_stats(1, P(dz), P(wt), P(x1), P(fmean), P(fstd), P(mask), P(part), B, H, W, st))


def fused():
    L.check(lib.sed_conv3x3_bwd_fused_c1(1, P(x1), P(fmean), P(fstd), P(w1), P(sc1), P(sh1), P(dy), P(z2), P(sc2), P(sh2), P(ca), P(cb), P(cc), 2,
                                         P(wt), P(mask), P(part), P(dwp), P(ws), B, H, W, C, P(dw), C, C, st))


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


res = {}
for r in range(rounds):
    for name, fn in (("b0 unfused (wgrad_c1 + dgrad_c1_stats)", unfused), ("b0 fused", fused)):
        res.setdefault(name, []).append(timeit(fn))
for k, v in res.items():
    v = sorted(v)
    print(f"{k:42s} median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}")
torch.cuda.synchronize()
fused()
torch.cuda.synchronize()
print(f"checksums: dW {float(dw.double().abs().sum()):.9e}  [A; sum g] {float(part.double().abs().sum()):.9e}")
