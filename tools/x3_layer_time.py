#!/usr/bin/env python3
"""Per-launch times of the split-operand kernels (csrc/sed_conv_x3.hip, dtype SED_F32H3) at the bench geometry (B = 32 x 60 s), through the
C ABI: forward, data gradient (ReLU-backward epilogue) and weight gradient of each 3x3 layer of blocks 0-3.
usage: x3_layer_time.py [layer substring] [iters]        (SED_DBG=16 with a `make STAMPS=1` build prints the in-kernel phase stamps)"""
import sys

import torch

sys.path.insert(0, ".")
import sed_amd  # noqa: E402

L = sed_amd._lib
lib = L.lib()
P = L.ptr
H3 = L.SED_F32H3
pat = sys.argv[1] if len(sys.argv) > 1 else ""
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
B = 32
LAYERS = [("b0c2", 6001, 64, 32, 32, 2, True), ("b1c1", 3000, 32, 32, 64, 1, False), ("b1c2", 3000, 32, 64, 64, 2, True),
          ("b2c1", 1500, 16, 64, 128, 1, False), ("b2c2", 1500, 16, 128, 128, 2, True), ("b3c1", 750, 8, 128, 128, 1, False),
          ("b3c2", 750, 8, 128, 128, 1, True)]


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, H, W, Ci, Co, pool, conv2 in LAYERS:
    if pat not in name:
        continue
    x = torch.randn(B, H, W, Ci, device=dev)
    z = torch.randn(B, H, W, Co, device=dev)
    g = torch.randn(B, H, W, Co, device=dev) * 2.0 ** -20
    dy = torch.randn(B, H // pool, W // pool, Co, device=dev) * 2.0 ** -20
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
    sc_i, sh_i = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.1
    sc_o, sh_o = torch.rand(Co, device=dev) + 0.5, torch.randn(Co, device=dev) * 0.1
    mean, invstd = torch.randn(Ci, device=dev) * 0.1, torch.rand(Ci, device=dev) + 0.5
    ca, cb, cc = torch.randn(Co, device=dev), torch.randn(Co, device=dev) * 1e-7, torch.randn(Co, device=dev) * 1e-7
    wp, wpt = torch.zeros(9 * Ci * Co, device=dev), torch.zeros(9 * Ci * Co, device=dev)
    L.check(lib.sed_pack_conv_weight(H3, P(w), P(wp), Co, Ci, Co, Ci, 0, st))
    L.check(lib.sed_pack_conv_weight(H3, P(w), P(wpt), Co, Ci, Co, Ci, 1, st))
    nparts = lib.sed_conv_nparts(B, H, W)
    out = torch.empty(B, H, W, Co, device=dev)
    gin = torch.empty(B, H, W, Ci, device=dev)
    part = torch.empty(nparts, 2, max(Ci, Co), device=dev)
    dzo = torch.empty(B, H, W, Co, device=dev)
    dwp, dw = torch.empty(9 * Ci * Co, device=dev), torch.empty(Co, Ci, 3, 3, device=dev)
    ws = torch.empty(lib.sed_conv_wgrad_ws_floats(B, H, W, Ci, Co), device=dev)
    dtg = H3 | (20 << 8)
    pro = 1 if conv2 else 0

    def fwd():
        L.check(lib.sed_conv3x3_fwd(H3, pro, 1, P(x), P(sc_i) if pro else None, P(sh_i) if pro else None, P(wp), P(out), None, None, None, None,
                                    None, P(part), B, H, W, Ci, Co, st))

    def dgrad():
        L.check(lib.sed_conv3x3_fwd(dtg, 0, 2, P(g), None, None, P(wpt), P(gin), P(x), P(sc_i), P(sh_i), P(mean), P(invstd), P(part), B, H, W, Co, Ci, st))

    def wgrad():
        if conv2:
            L.check(lib.sed_conv3x3_wgrad_fused_u(dtg, 1, P(x), P(sc_i), P(sh_i), 1, P(dy), P(z), P(sc_o), P(sh_o), P(ca), P(cb), P(cc), pool, P(dzo),
                                                  P(dwp), P(ws), B, H, W, Ci, Co, P(dw), Co, Ci, st))
        else:
            L.check(lib.sed_conv3x3_wgrad_fused_u(dtg, 0, P(x), None, None, 2, P(g), P(z), None, None, P(ca), P(cb), P(cc), 1, P(dzo), P(dwp), P(ws),
                                                  B, H, W, Ci, Co, P(dw), Co, Ci, st))

    gf = 2 * B * H * W * 9 * Ci * Co / 1e9
    tf, td, tw = timeit(fwd), timeit(dgrad), timeit(wgrad)
    print(f"{name} {Ci}->{Co} H{H} W{W}: fwd {tf:.3f} ms ({3 * gf / tf:.0f} TF/s of 16-bit MFMA)  dgrad {td:.3f} ms ({3 * gf / td:.0f})  "
          f"wgrad(+reduce) {tw:.3f} ms ({3 * gf / tw:.0f})", flush=True)
    del x, z, g, dy, out, gin, dzo, ws
