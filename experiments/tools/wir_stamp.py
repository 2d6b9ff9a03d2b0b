#!/usr/bin/env python3
"""In-kernel phase stamps of the resident-weight conv kernel (SED_DBG=16): one launch of `fwd PRO_NONE EPI_STORE`."""
import os, sys
import torch
sys.path.insert(0, ".")
os.environ["SED_CONV_KERNEL"] = "r"
import sed_amd
L = sed_amd._lib; lib = L.lib(); P = L.ptr
B, H, W, Cin, Cout = [int(v) for v in sys.argv[1:6]]
epi = int(sys.argv[6]) if len(sys.argv) > 6 else 0
bf = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, H, W, Cin, device="cuda").to(bf)
out = torch.empty(B, H, W, Cout, device="cuda", dtype=bf)
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
wpack = torch.empty(9 * Cin * Cout, device="cuda", dtype=bf)
part = torch.empty(lib.sed_conv_nparts(B, H, W) * 2 * Cout, device="cuda")
L.check(lib.sed_pack_conv_weight(1, P(w), P(wpack), Cout, Cin, Cout, Cin, 0, st))
for _ in range(3):
    L.check(lib.sed_conv3x3_fwd(1, 0, epi, P(x), None, None, P(wpack), P(out), None, None, None, None, None, P(part), B, H, W, Cin, Cout, st))
torch.cuda.synchronize()
os.environ["SED_DBG"] = "16"
lib.sed_config_reload()
L.check(lib.sed_conv3x3_fwd(1, 0, epi, P(x), None, None, P(wpack), P(out), None, None, None, None, None, P(part), B, H, W, Cin, Cout, st))
torch.cuda.synchronize()
