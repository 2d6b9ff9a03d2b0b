#!/usr/bin/env python3
"""Ablation timings (SED_DBG switches: 1 no stores, 2 no MFMA loop, 4 no conv1 rebuild (forward), 8 no global loads) of block 0's
three C1-mode kernels at the BENCH geometry (B x 6001 x 64, 32 -> 32 channels)."""
import os, sys
import torch
sys.path.insert(0, ".")
import sed_amd
L = sed_amd._lib; lib = L.lib(); P = L.ptr
bf = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
B, H, W = 32, 6001, 64
x = torch.randn(B, H, W, device="cuda")
w1 = torch.randn(32, 1, 3, 3, device="cuda") * 0.3
w2 = torch.randn(32, 32, 3, 3, device="cuda") * 0.05
wpack = torch.empty(9 * 32 * 32, device="cuda", dtype=bf); wpack_t = torch.empty_like(wpack)
L.check(lib.sed_pack_conv_weight(1, P(w2), P(wpack), 32, 32, 32, 32, 0, st))
L.check(lib.sed_pack_conv_weight(1, P(w2), P(wpack_t), 32, 32, 32, 32, 1, st))
sc, sh = torch.rand(32, device="cuda") + 0.5, torch.randn(32, device="cuda") * 0.1
ca, cb, cc = torch.randn(32, device="cuda"), torch.randn(32, device="cuda") * 0.1, torch.randn(32, device="cuda") * 0.1
z = torch.randn(B, H, W, 32, device="cuda").to(bf)
dy = torch.randn(B, H // 2, W // 2, 32, device="cuda").to(bf)
dz = torch.empty(B, H, W, 32, device="cuda", dtype=bf)
part = torch.empty(lib.sed_conv_nparts(B, H, W) * 2 * 32, device="cuda")
mask = torch.zeros(B, H, W, 2, device="cuda", dtype=torch.int16)
dwp = torch.empty(9 * 32 * 32, device="cuda")
ws = torch.empty(lib.sed_conv_wgrad_ws_floats(B, H, W, 32, 32), device="cuda")
a10 = torch.empty(lib.sed_conv_dgrad_c1_nparts() * 10 * 32, device="cuda")
calls = {
    "fwd_c1": lambda: L.check(lib.sed_conv3x3_fwd_c1(1, 1, P(x), None, None, P(w1), P(sc), P(sh), P(wpack), P(z), P(part), P(mask), B, H, W, 32, st)),
    "wgrad_c1": lambda: L.check(lib.sed_conv3x3_wgrad_fused_c1(1, P(x), None, None, P(w1), P(sc), P(sh), P(dy), P(z), P(sc), P(sh), P(ca), P(cb), P(cc), 2, P(dz), P(dwp), P(ws), B, H, W, 32, st)),
    "dgrad_c1": lambda: L.check(lib.sed_conv3x3_dgrad_c1_stats(1, P(dz), P(wpack_t), P(x), None, None, P(mask), P(a10), B, H, W, st)),
}
for name, call in calls.items():
    row = []
    for d in (0, 1, 2, 8, 3, 11) + ((4, 6, 7) if name == "fwd_c1" else ()):
        os.environ["SED_DBG"] = str(d); lib.sed_config_reload()
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        row.append(f"dbg{d}={e0.elapsed_time(e1) / 10:.3f}")
    print(name, " ".join(row))
