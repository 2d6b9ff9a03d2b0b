#!/usr/bin/env python3
"""Times sed_logmel_fwd on the BENCH workload (B x 60 s @ 32 kHz) for both front-end kernels (1 = one wave per frame, default = 8-frame batches / fp32 MFMA mel) (GPU box only)."""
import os, sys
import torch
sys.path.insert(0, ".")
import sed_amd
pp = __import__("importlib").import_module("soundeventdetection-pytorch_amd.dataset.spectogram.preprocess")
sc = __import__("importlib").import_module("soundeventdetection-pytorch_amd.dataset.spectogram.spectogram_configs")
lib = sed_amd._lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
c = sc.BENCH
w = (torch.randn(B, 60 * c.working_sample_rate) * 0.1).clamp_(-1, 1).cuda()
fe = pp.LogMelFrontEnd(c, "cuda")
outs = {}
for k in ("1", "0"):
    os.environ["SED_FE_KERNEL"] = k
    lib.sed_config_reload()
    out = fe(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fe(w, out=out)
    e1.record()
    torch.cuda.synchronize()
    outs[k] = out.clone()
    print(f"SED_FE_KERNEL={k}: {e0.elapsed_time(e1) / 20:.4f} ms   ({B * 6001 / (e0.elapsed_time(e1) / 20) / 1e3:.1f} M frames/s)")
print("max |default - kernel 1| dB:", (outs["0"] - outs["1"]).abs().max().item())
