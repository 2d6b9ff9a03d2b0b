#!/usr/bin/env python3
"""In-kernel phase stamps (SED_DBG=16) of the producer/consumer conv kernel on one layer shape, or (arg "c1") of block 0's
C1-mode forward through the model engine."""
import os, sys
import torch
sys.path.insert(0, ".")
import sed_amd
L = sed_amd._lib; lib = L.lib(); P = L.ptr
bf = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
if sys.argv[1] == "c1":
    B, H, W = 32, 6001, 64
    x = torch.randn(B, H, W, device="cuda")
    w1 = torch.randn(32, 1, 3, 3, device="cuda") * 0.3
    w2 = torch.randn(32, 32, 3, 3, device="cuda") * 0.05
    wpack = torch.empty(9 * 32 * 32, device="cuda", dtype=bf)
    L.check(lib.sed_pack_conv_weight(1, P(w2), P(wpack), 32, 32, 32, 32, 0, st))
    sc, sh = torch.rand(32, device="cuda") + 0.5, torch.randn(32, device="cuda") * 0.1
    z = torch.empty(B, H, W, 32, device="cuda", dtype=bf)
    part = torch.empty(lib.sed_conv_nparts(B, H, W) * 2 * 32, device="cuda")
    mask = torch.empty(B, H, W, 2, device="cuda", dtype=torch.int16)
    call = lambda: L.check(lib.sed_conv3x3_fwd_c1(1, 1, P(x), None, None, P(w1), P(sc), P(sh), P(wpack), P(z), P(part), None if os.environ.get("PC_STAMP_NOMASK") == "1" else P(mask), B, H, W, 32, st))      # PC_STAMP_NOMASK=1: the round-5 form (no mask built / stored)
else:
    B, H, W, Cin, Cout = [int(v) for v in sys.argv[1:6]]
    pro = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    x = torch.randn(B, H, W, Cin, device="cuda").to(bf)
    out = torch.empty(B, H, W, Cout, device="cuda", dtype=bf)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wpack = torch.empty(9 * Cin * Cout, device="cuda", dtype=bf)
    part = torch.empty(lib.sed_conv_nparts(B, H, W) * 2 * Cout, device="cuda")
    sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
    L.check(lib.sed_pack_conv_weight(1, P(w), P(wpack), Cout, Cin, Cout, Cin, 0, st))
    call = lambda: L.check(lib.sed_conv3x3_fwd(1, pro, 1, P(x), P(sc) if pro else None, P(sh) if pro else None, P(wpack), P(out), None, None, None, None, None, P(part), B, H, W, Cin, Cout, st))
for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    call()
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / 10:.4f} ms per launch")
os.environ["SED_DBG"] = "16"
lib.sed_config_reload()
call()
torch.cuda.synchronize()
