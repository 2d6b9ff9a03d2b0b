#!/usr/bin/env python3
"""sed_bn_relu_pool_cnt_fwd at the bench geometry of blocks 0 / 1 (B = 32): the generic row kernel (SED_POOL_PAIR=0) against the
pair-lane kernel (thread = input column, partner column by DPP), interleaved in one process; outputs compared.  usage: ab_pool_pair.py [rounds]"""
import os
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
g = torch.Generator(device=dev).manual_seed(3)
res = {}
for name, (B, H, W, C) in {"b0 6001x64x32": (32, 6001, 64, 32), "b1 3000x32x64": (32, 3000, 32, 64)}.items():
    z = torch.randn(B, H, W, C, device=dev, generator=g).to(bf)
    sc, sh = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.3
    outs = {}
    for mode in ("0", "1"):
        os.environ["SED_POOL_PAIR"] = mode
        lib.sed_config_reload()
        y = torch.full((B, H // 2, W // 2, C), 7.0, device=dev, dtype=bf)
        cnt = torch.full((B, H // 2, W // 2, C), 9, device=dev, dtype=torch.uint8)
        call = lambda: L.check(lib.sed_bn_relu_pool_cnt_fwd(1, P(z), P(sc), P(sh), P(y), P(cnt), B, H, W, C, st))
        call()
        torch.cuda.synchronize()
        outs[mode] = (y.clone(), cnt.clone())
    d = (outs["0"][0].float() - outs["1"][0].float()).abs()
    rel = float((d / outs["0"][0].float().abs().clamp_min(1e-3)).max())
    print(f"{name}: counts identical {torch.equal(outs['0'][1], outs['1'][1])}   y max rel diff {rel:.2e} (one bf16 ulp = 7.8e-3)   differing elements {int((d > 0).sum())} of {d.numel()}")
    for r in range(rounds):
        for mode in ("0", "1"):
            os.environ["SED_POOL_PAIR"] = mode
            lib.sed_config_reload()
            call = lambda: L.check(lib.sed_bn_relu_pool_cnt_fwd(1, P(z), P(sc), P(sh), P(y), P(cnt), B, H, W, C, st))
            call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault((name, mode), []).append(e0.elapsed_time(e1) / 10)
for (name, mode), v in res.items():
    v = sorted(v)
    print(f"{name}  SED_POOL_PAIR={mode}   median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}")
