#!/usr/bin/env python3
"""Micro-benchmark of single conv-type kernels through the C ABI on random data (GPU box only).
Prints ms per launch for the weight-gradient variants and the forward / data-gradient kernels of one
layer shape.   usage: bench_layer.py B H W Cin Cout [iters]"""
import sys

import torch

sys.path.insert(0, ".")
import sed_amd  # noqa: E402

L = sed_amd._lib
lib = L.lib()
B, H, W, Cin, Cout = [int(v) for v in sys.argv[1:6]]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
x = torch.randn(B, H, W, Cin, device=dev).to(bf)
z = torch.randn(B, H, W, Cout, device=dev).to(bf)
dz = torch.randn(B, H, W, Cout, device=dev).to(bf)
dy = torch.randn(B, H // 2, W // 2, Cout, device=dev).to(bf)
out = torch.empty(B, H, W, Cout, device=dev, dtype=bf)
outc = torch.empty(B, H, W, Cin, device=dev, dtype=bf)
sc_i, sh_i = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
sc_o, sh_o = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
ca, cb, cc = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev) * 0.1, torch.randn(Cout, device=dev) * 0.1
mean, invstd = torch.randn(Cin, device=dev) * 0.1, torch.rand(Cin, device=dev) + 0.5
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
wpack = torch.empty(9 * Cin * Cout, device=dev, dtype=bf)
wpack_t = torch.empty(9 * Cin * Cout, device=dev, dtype=bf)
P = L.ptr
L.check(lib.sed_pack_conv_weight(1, L.ptr(w), L.ptr(wpack), Cout, Cin, Cout, Cin, 0, st))
L.check(lib.sed_pack_conv_weight(1, L.ptr(w), L.ptr(wpack_t), Cout, Cin, Cout, Cin, 1, st))
dwp = torch.empty(9 * Cin * Cout, device=dev)
ws = torch.empty(lib.sed_conv_wgrad_ws_floats(B, H, W, Cin, Cout), device=dev)
part = torch.empty(lib.sed_conv_nparts(B, H, W) * 2 * max(Cin, Cout), device=dev)


def timeit(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    gf = 2.0 * 9 * Cin * Cout * B * H * W / 1e9
    print(f"{name:44s} {ms:8.3f} ms   {gf / ms:8.1f} TFLOP/s-equivalent")


P = L.ptr
print(f"layer B={B} H={H} W={W} {Cin}->{Cout}  ({2.0 * 9 * Cin * Cout * B * H * W / 1e9:.1f} GFLOP)")
import os
if os.environ.get("ONLY_WGRAD"):
    ws = torch.empty(2048 * 9 * Cin * Cout + 1024, device=dev)      # enough for any block count below
    for nb in (256, 512, 768, 1024, 1536, 2048):
        os.environ["SED_WGRAD_BLOCKS"] = str(nb)
        lib.sed_config_reload()
        timeit(f"wgrad PRO_NONE DZ_GIVEN blocks={nb}", lambda: L.check(lib.sed_conv3x3_wgrad(1, 0, P(x), None, None, P(dz), P(dwp), P(ws), B, H, W, Cin, Cout, st)))
    sys.exit(0)
timeit("wgrad  PRO_NONE   DZ_GIVEN", lambda: L.check(lib.sed_conv3x3_wgrad(1, 0, P(x), None, None, P(dz), P(dwp), P(ws), B, H, W, Cin, Cout, st)))
timeit("wgrad  PRO_BNRELU DZ_GIVEN", lambda: L.check(lib.sed_conv3x3_wgrad(1, 1, P(x), P(sc_i), P(sh_i), P(dz), P(dwp), P(ws), B, H, W, Cin, Cout, st)))
timeit("wgrad  PRO_NONE   DZ_BN   (+dz_out)", lambda: L.check(lib.sed_conv3x3_wgrad_fused(1, 0, P(x), None, None, 2, P(dz), P(z), None, None, P(ca), P(cb), P(cc), 1, P(out), P(dwp), P(ws), B, H, W, Cin, Cout, st)))
timeit("wgrad  PRO_BNRELU DZ_POOL (+dz_out)", lambda: L.check(lib.sed_conv3x3_wgrad_fused(1, 1, P(x), P(sc_i), P(sh_i), 1, P(dy), P(z), P(sc_o), P(sh_o), P(ca), P(cb), P(cc), 2, P(out), P(dwp), P(ws), B, H, W, Cin, Cout, st)))
timeit("wgrad  PRO_BNRELU DZ_POOL (no dz_out)", lambda: L.check(lib.sed_conv3x3_wgrad_fused(1, 1, P(x), P(sc_i), P(sh_i), 1, P(dy), P(z), P(sc_o), P(sh_o), P(ca), P(cb), P(cc), 2, None, P(dwp), P(ws), B, H, W, Cin, Cout, st)))
timeit("fwd    PRO_NONE   EPI_STATS", lambda: L.check(lib.sed_conv3x3_fwd(1, 0, 1, P(x), None, None, P(wpack), P(out), None, None, None, None, None, P(part), B, H, W, Cin, Cout, st)))
timeit("fwd    PRO_BNRELU EPI_STATS", lambda: L.check(lib.sed_conv3x3_fwd(1, 1, 1, P(x), P(sc_i), P(sh_i), P(wpack), P(out), None, None, None, None, None, P(part), B, H, W, Cin, Cout, st)))
timeit("fwd    PRO_NONE   EPI_STORE", lambda: L.check(lib.sed_conv3x3_fwd(1, 0, 0, P(x), None, None, P(wpack), P(out), None, None, None, None, None, None, B, H, W, Cin, Cout, st)))
timeit("dgrad  PRO_NONE   EPI_RELUBWD", lambda: L.check(lib.sed_conv3x3_fwd(1, 0, 2, P(dz), None, None, P(wpack_t), P(outc), P(x), P(sc_i), P(sh_i), P(mean), P(invstd), P(part), B, H, W, Cout, Cin, st)))
timeit("dgrad  PRO_NONE   EPI_STORE", lambda: L.check(lib.sed_conv3x3_fwd(1, 0, 0, P(dz), None, None, P(wpack_t), P(outc), None, None, None, None, None, None, B, H, W, Cout, Cin, st)))
