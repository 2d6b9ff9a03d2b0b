"""Kernel-level A/B on the MI355X: the producer/consumer convolution kernels (sed_conv_pc.hip, sed_wgrad.hip) against
the previous-generation kernels (conv_igemm_kernel / conv_wgrad2_kernel, selected with SED_CONV_KERNEL=lds /
SED_WGRAD_KERNEL=2) through the C ABI, on the same random bf16 operands.  Both accumulate the same bf16 products in
fp32, so outputs agree to summation order (one bf16 ulp after rounding); shapes exercise ragged heights (last tile
partly outside the image), single images, the pooling floor and every prologue / epilogue / dz mode of the path."""
import importlib
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # B, H, W, Cin, Cout
    (2, 37, 64, 32, 32),
    (1, 9, 64, 32, 32),
    (3, 50, 32, 64, 64),
    (2, 33, 32, 32, 64),
    (2, 41, 16, 128, 128),
    (2, 29, 16, 64, 128),
    (2, 70, 16, 128, 64),
]


@pytest.fixture(scope="module")
def L():
    sed = importlib.import_module("soundeventdetection-pytorch_amd")
    return sed._lib


def _close_bf16(a, b, what):
    a, b = a.float(), b.float()
    tol = 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 2e-3          # 2 bf16 ulps + accumulation noise near 0
    bad = ((a - b).abs() > tol).float().mean().item()
    assert bad == 0.0, f"{what}: {100 * bad:.3f}% of the elements differ by more than 2 bf16 ulps"


def _both(monkeypatch, fn):
    out = []
    for conv, wg in (("p", "3"), ("lds", "2")):
        monkeypatch.setenv("SED_CONV_KERNEL", conv)
        monkeypatch.setenv("SED_WGRAD_KERNEL", wg)
        out.append(fn())
        torch.cuda.synchronize()
    monkeypatch.delenv("SED_CONV_KERNEL")
    monkeypatch.delenv("SED_WGRAD_KERNEL")
    return out


@pytest.mark.parametrize("B,H,W,Cin,Cout", SHAPES)
def test_forward_and_data_gradient_kernels(L, monkeypatch, B, H, W, Cin, Cout):
    lib, P, dev, bf = L.lib(), L.ptr, "cuda", torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + H)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g).to(bf)
    dz = torch.randn(B, H, W, Cout, device=dev, generator=g).to(bf)
    sc_i, sh_i = torch.rand(Cin, device=dev, generator=g) + 0.5, torch.randn(Cin, device=dev, generator=g) * 0.3
    mean, invstd = torch.randn(Cin, device=dev, generator=g) * 0.1, torch.rand(Cin, device=dev, generator=g) + 0.5
    w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) * 0.05
    wpack = torch.empty(9 * Cin * Cout, device=dev, dtype=bf)
    wpack_t = torch.empty(9 * Cin * Cout, device=dev, dtype=bf)
    L.check(lib.sed_pack_conv_weight(1, P(w), P(wpack), Cout, Cin, Cout, Cin, 0, st))
    L.check(lib.sed_pack_conv_weight(1, P(w), P(wpack_t), Cout, Cin, Cout, Cin, 1, st))
    nparts = lib.sed_conv_nparts(B, H, W)

    def fwd(pro, epi):
        def run():
            out = torch.full((B, H, W, Cout), 7.0, device=dev, dtype=bf)
            part = torch.full((nparts, 2, Cout), 3.0, device=dev)
            L.check(lib.sed_conv3x3_fwd(1, pro, epi, P(x), P(sc_i) if pro else None, P(sh_i) if pro else None, P(wpack), P(out), None,
                                        None, None, None, None, P(part) if epi else None, B, H, W, Cin, Cout, st))
            return out, part.sum(0)
        return run

    for pro, epi in ((1, 1), (0, 1), (0, 0), (1, 0)):
        (za, sa), (zb, sb) = _both(monkeypatch, fwd(pro, epi))
        _close_bf16(za, zb, f"fwd pro={pro} epi={epi}")
        if epi:
            torch.testing.assert_close(sa, sb, rtol=2e-3, atol=2e-2 * (B * H * W) ** 0.5)

    def dgrad():
        out = torch.full((B, H, W, Cin), 7.0, device=dev, dtype=bf)
        part = torch.full((nparts, 2, Cin), 3.0, device=dev)
        L.check(lib.sed_conv3x3_fwd(1, 0, 2, P(dz), None, None, P(wpack_t), P(out), P(x), P(sc_i), P(sh_i), P(mean), P(invstd), P(part),
                                    B, H, W, Cout, Cin, st))
        return out, part.sum(0)

    (ga, pa), (gb, pb) = _both(monkeypatch, dgrad)
    # the ReLU mask is taken from the same stored reference in both kernels: identical decisions
    assert torch.equal(ga == 0, gb == 0)
    _close_bf16(ga, gb, "dgrad RELUBWD")
    torch.testing.assert_close(pa, pb, rtol=5e-3, atol=5e-2 * (B * H * W) ** 0.5)


@pytest.mark.parametrize("B,H,W,Cin,Cout", SHAPES)
def test_weight_gradient_kernels(L, monkeypatch, B, H, W, Cin, Cout):
    lib, P, dev, bf = L.lib(), L.ptr, "cuda", torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(B * 77 + H)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g).to(bf)
    z = torch.randn(B, H, W, Cout, device=dev, generator=g).to(bf)
    dz = torch.randn(B, H, W, Cout, device=dev, generator=g).to(bf)
    dy = torch.randn(B, H // 2, W // 2, Cout, device=dev, generator=g).to(bf)
    sc_i, sh_i = torch.rand(Cin, device=dev, generator=g) + 0.5, torch.randn(Cin, device=dev, generator=g) * 0.3
    sc_o, sh_o = torch.rand(Cout, device=dev, generator=g) + 0.5, torch.randn(Cout, device=dev, generator=g) * 0.3
    ca, cb, cc = (torch.randn(Cout, device=dev, generator=g) * s for s in (1.0, 0.1, 0.1))
    n = 9 * Cin * Cout

    def run(mode):
        def f():
            ws = torch.empty(lib.sed_conv_wgrad_ws_floats(B, H, W, Cin, Cout), device=dev)
            dwp = torch.full((n,), 5.0, device=dev)
            out = torch.full((B, H, W, Cout), 7.0, device=dev, dtype=bf)
            if mode == "plain":
                L.check(lib.sed_conv3x3_wgrad(1, 0, P(x), None, None, P(dz), P(dwp), P(ws), B, H, W, Cin, Cout, st))
            elif mode == "plain_pro":
                L.check(lib.sed_conv3x3_wgrad(1, 1, P(x), P(sc_i), P(sh_i), P(dz), P(dwp), P(ws), B, H, W, Cin, Cout, st))
            elif mode == "bn":
                L.check(lib.sed_conv3x3_wgrad_fused(1, 0, P(x), None, None, 2, P(dz), P(z), None, None, P(ca), P(cb), P(cc), 1, P(out),
                                                    P(dwp), P(ws), B, H, W, Cin, Cout, st))
            else:
                L.check(lib.sed_conv3x3_wgrad_fused(1, 1, P(x), P(sc_i), P(sh_i), 1, P(dy), P(z), P(sc_o), P(sh_o), P(ca), P(cb), P(cc),
                                                    2, P(out), P(dwp), P(ws), B, H, W, Cin, Cout, st))
            return dwp, out
        return f

    for mode in ("plain", "plain_pro", "bn", "pool"):
        (da, oa), (db, ob) = _both(monkeypatch, run(mode))
        scale = float(db.abs().max()) + 1e-6
        assert float((da - db).abs().max()) / scale < 2e-3, mode           # fp32 sums of identical bf16 products
        if mode in ("bn", "pool"):
            _close_bf16(oa, ob, f"dz_out {mode}")
