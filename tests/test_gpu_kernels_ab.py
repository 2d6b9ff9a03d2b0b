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
    (2, 45, 8, 128, 128),
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


def _reload():
    importlib.import_module("soundeventdetection-pytorch_amd")._lib.lib().sed_config_reload()


def _both(monkeypatch, fn, first="p"):
    out = []
    for conv, wg in ((first, "3"), ("lds", "2")):
        monkeypatch.setenv("SED_CONV_KERNEL", conv)
        monkeypatch.setenv("SED_WGRAD_KERNEL", wg)
        _reload()                    # (the library caches its SED_* knobs: sed_config_reload is the test hook)
        out.append(fn())
        torch.cuda.synchronize()
    monkeypatch.delenv("SED_CONV_KERNEL")
    monkeypatch.delenv("SED_WGRAD_KERNEL")
    _reload()
    return out


@pytest.mark.parametrize("B,H,W,Cin,Cout", SHAPES)
def test_forward_and_data_gradient_kernels(L, monkeypatch, B, H, W, Cin, Cout, first="p"):
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
        (za, sa), (zb, sb) = _both(monkeypatch, fwd(pro, epi), first)
        _close_bf16(za, zb, f"fwd pro={pro} epi={epi}")
        if epi:
            torch.testing.assert_close(sa, sb, rtol=2e-3, atol=2e-2 * (B * H * W) ** 0.5)

    def dgrad():
        out = torch.full((B, H, W, Cin), 7.0, device=dev, dtype=bf)
        part = torch.full((nparts, 2, Cin), 3.0, device=dev)
        L.check(lib.sed_conv3x3_fwd(1, 0, 2, P(dz), None, None, P(wpack_t), P(out), P(x), P(sc_i), P(sh_i), P(mean), P(invstd), P(part),
                                    B, H, W, Cout, Cin, st))
        return out, part.sum(0)

    (ga, pa), (gb, pb) = _both(monkeypatch, dgrad, first)
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


@pytest.mark.parametrize("th", ["8", "4"])
@pytest.mark.parametrize("B,H", [(2, 37), (1, 9), (3, 41), (1, 6), (2, 64)])
def test_fused_first_block_data_gradient(L, monkeypatch, B, H, th):
    """sed_conv3x3_dgrad_c1_stats (csrc/sed_dgrad_c1.hip: g never written, A = sum_px g (x) patch and sum g contracted on the
    matrix pipe from the gated accumulators) against (a) a torch fp32 restatement on the same bf16 operands and (b) the
    unfused kernels it replaces (sed_conv3x3_dgrad_c1 -> g -> sed_conv3x3_c1_wgrad)."""
    import torch.nn.functional as F
    monkeypatch.setenv("SED_DGRAD_TH", th)
    _reload()
    lib, P, dev, bf = L.lib(), L.ptr, "cuda", torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    W, C = 64, 32
    g = torch.Generator(device="cuda").manual_seed(B * 131 + H)
    dz = torch.randn(B, H, W, C, device=dev, generator=g).to(bf)
    w2 = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05
    x1 = torch.randn(B, H, W, device=dev, generator=g) * 3.0 + 1.0
    fmean = torch.randn(W, device=dev, generator=g)
    fstd = torch.rand(W, device=dev, generator=g) + 0.5
    mask = torch.randint(0, 65536, (B, H, W, 2), device=dev, generator=g, dtype=torch.int32).to(torch.int16)
    wpack_t = torch.empty(9 * C * C, device=dev, dtype=bf)
    L.check(lib.sed_pack_conv_weight(1, P(w2), P(wpack_t), C, C, C, C, 1, st))
    nparts = lib.sed_conv_dgrad_c1_nparts()
    part = torch.full((nparts, 10, C), 9.0, device=dev)
    L.check(lib.sed_conv3x3_dgrad_c1_stats(1, P(dz), P(wpack_t), P(x1), P(fmean), P(fstd), P(mask), P(part), B, H, W, st))
    torch.cuda.synchronize()
    got = part.double().sum(0).cpu()                                     # [10][32]
    # the test hook stores the gated g as well; the statistics of that launch are the same bits
    part_g = torch.full((nparts, 10, C), 9.0, device=dev)
    g_new = torch.full((B, H, W, C), 7.0, device=dev, dtype=bf)
    L.check(lib.sed_conv3x3_dgrad_c1_stats_g(1, P(dz), P(wpack_t), P(x1), P(fmean), P(fstd), P(mask), P(part_g), P(g_new), B, H, W, st))
    torch.cuda.synchronize()
    assert torch.equal(part, part_g)

    # (a) torch restatement (CPU fp32/fp64 on the bf16-rounded operands)
    dzc = dz.float().cpu().permute(0, 3, 1, 2)
    w2c = w2.to(bf).float().cpu()
    gpre = F.conv_transpose2d(dzc, w2c, padding=1)                        # [B][c1][H][W]
    mk = mask.cpu().to(torch.int32) & 0xFFFF
    c = torch.arange(C)
    half, bit = (c >> 2) & 1, (c & 3) + 4 * (c >> 3)
    on = ((mk[..., half] >> bit) & 1).permute(0, 3, 1, 2).float()         # [B][c][H][W]
    gg = (gpre * on).to(bf).double()
    xz = ((x1 - fmean) / fstd).to(bf).double().cpu()
    xp = F.pad(xz, (1, 1, 1, 1))
    ref = torch.zeros(10, C, dtype=torch.float64)
    for k in range(9):
        ti, tj = divmod(k, 3)
        ref[k] = (gg * xp[:, None, ti:ti + H, tj:tj + W]).sum(dim=(0, 2, 3))
    ref[9] = gg.sum(dim=(0, 2, 3))
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) / scale < 2e-4, float((got - ref).abs().max()) / scale

    # (b) the unfused kernels
    gbuf = torch.empty(B, H, W, C, device=dev, dtype=bf)
    np2 = lib.sed_conv_nparts(B, H, W)
    part2 = torch.zeros(np2, 2, C, device=dev)
    L.check(lib.sed_conv3x3_dgrad_c1(1, P(dz), P(wpack_t), P(gbuf), P(mask), P(part2), B, H, W, C, st))
    np1 = lib.sed_conv_c1_nparts(B, H, W)
    ws = torch.zeros(np1, 9, C, device=dev)
    L.check(lib.sed_conv3x3_c1_wgrad(1, P(x1), P(fmean), P(fstd), P(gbuf), P(ws), B, H, W, C, st))
    torch.cuda.synchronize()
    assert torch.equal(g_new, gbuf)                                       # same bf16 products, same fp32 accumulation order
    old = torch.cat([ws.double().sum(0), part2[:, 0].double().sum(0)[None]]).cpu()
    assert float((got - old).abs().max()) / scale < 6e-3                  # the unfused pair keeps x in fp32, the fused one rounds it to bf16


def test_batched_weight_pack_matches_single_launches(L):
    """sed_pack_conv_weights_batch (one launch, device descriptor table) against one sed_pack_conv_weight launch per operand."""
    lib, P, dev, bf = L.lib(), L.ptr, "cuda", torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(5)
    layers = [(32, 32), (64, 32), (64, 64), (128, 64), (40, 24)]            # (Cout, Cin); the last one exercises channel padding
    rows, blk, refs, outs = [], 0, [], []
    pad = lambda c: (c + 31) // 32 * 32
    for co, ci in layers:
        w = torch.randn(co, ci, 3, 3, device=dev, generator=g)
        for tf in (0, 1):
            cop, cip = pad(co), pad(ci)
            ref = torch.zeros(9 * cop * cip, device=dev, dtype=bf)
            L.check(lib.sed_pack_conv_weight(1, P(w), P(ref), co, ci, cop, cip, tf, st))
            out = torch.full((9 * cop * cip,), 3.0, device=dev, dtype=bf)
            pop, pip_ = (cip, cop) if tf else (cop, cip)
            rows.append([w.data_ptr(), out.data_ptr(), co, ci, pop, pip_, tf, blk])
            blk += (pip_ * 9 * pop + 1023) // 1024
            refs.append(ref); outs.append(out); outs.append(w)               # keep the weights alive
    desc = torch.tensor(rows, dtype=torch.int64).to(dev)
    L.check(lib.sed_pack_conv_weights_batch(1, P(desc), len(rows), blk, st))
    torch.cuda.synchronize()
    for i, ref in enumerate(refs):
        assert torch.equal(ref, outs[2 * i]), f"operand {i}"


def test_prefetching_front_end_matches_direct_calls():
    """PrefetchingFrontEnd (second stream, double buffer) returns the same features, in submission order, as direct calls."""
    sed = importlib.import_module("soundeventdetection-pytorch_amd")
    pp = importlib.import_module("soundeventdetection-pytorch_amd.dataset.spectogram.preprocess")
    sc = importlib.import_module("soundeventdetection-pytorch_amd.dataset.spectogram.spectogram_configs")
    fe = pp.LogMelFrontEnd(sc.BENCH, "cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    waves = [torch.randn(2, 32000, device="cuda", generator=g) * 0.1 for _ in range(4)]
    want = [fe(w).clone() for w in waves]
    pf = pp.PrefetchingFrontEnd(fe)
    pf.submit(waves[0])
    for i in range(4):
        x = pf.get()
        if i + 1 < 4:
            pf.submit(waves[i + 1])
        got = x.clone()
        pf.release()
        torch.cuda.synchronize()
        assert torch.equal(got, want[i]), i
    with pytest.raises(RuntimeError):
        pf.get()


@pytest.mark.parametrize("B,Lw", [(8, 31680), (16, 2048), (8, 1111)])
def test_m5_first_layer_on_the_matrix_pipe(L, monkeypatch, B, Lw):
    """sed_m5_mfma.hip (bf16 MFMA forward; weight gradient with the BatchNorm backward rebuilt on load) against the fp32 VALU
    kernels of sed_m5.hip on the same operands: forward within bf16 rounding of the 79-tap products, statistics and weight
    gradient relative to their scale."""
    lib, P, dev, bf = L.lib(), L.ptr, "cuda", torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(B + Lw)
    x = torch.randn(B, Lw, device=dev, generator=g) * 0.3
    w = torch.randn(64, 79, device=dev, generator=g) * 0.1
    L1, npt = lib.sed_m5_conv1_len(Lw), lib.sed_m5_conv1_nparts(B, Lw)

    def fwd(flag):
        monkeypatch.setenv("SED_M5_MFMA", flag)
        _reload()
        z = torch.full((B // 8, L1, 8, 64), 7.0, device=dev, dtype=bf)
        part = torch.full((npt, 2, 64), 3.0, device=dev)
        L.check(lib.sed_m5_conv1_fwd(1, P(x), P(w), P(z), P(part), B, Lw, st))
        torch.cuda.synchronize()
        return z, part.double().sum(0)

    (z1, s1), (z0, s0) = fwd("1"), fwd("0")
    monkeypatch.delenv("SED_M5_MFMA")
    _reload()
    # exact reference of the bf16-operand product: x and w rounded to bf16, fp32 accumulation
    xr, wr = x.to(bf).float(), w.to(bf).float()
    ref = torch.nn.functional.conv1d(xr[:, None].cpu(), wr[:, None].cpu(), stride=4, padding=39)        # [B][64][L1]
    got = z1.float().cpu().permute(0, 2, 3, 1).reshape(B, 64, L1)                                         # [(n, w)][c][t]
    assert float((got - ref).abs().max()) < 2.0 ** -7 * float(ref.abs().max()) + 1e-3
    assert float((z1.float() - z0.float()).abs().max()) < 0.05 * float(z0.float().abs().max())          # vs fp32-operand VALU kernel
    torch.testing.assert_close(s1, s0, rtol=2e-2, atol=2e-2 * float(s0.abs().max()))

    gg = torch.randn(B // 8, L1, 8, 64, device=dev, generator=g).to(bf)
    ca, cb, cc = (torch.randn(64, device=dev, generator=g) * s for s in (1.0, 0.2, 0.1))
    ws1 = torch.full((npt, 80, 64), 5.0, device=dev)
    L.check(lib.sed_m5_conv1_wgrad_fused(1, P(x), P(gg), P(z0), P(ca), P(cb), P(cc), P(ws1), B, Lw, st))
    dz = torch.empty_like(gg)
    L.check(lib.sed_bn_bwd_apply(1, P(gg), P(z0), P(ca), P(cb), P(cc), P(dz), (B // 8) * L1 * 8, 64, st))
    ws0 = torch.full((npt, 80, 64), 5.0, device=dev)
    L.check(lib.sed_m5_conv1_wgrad(1, P(x), P(dz), P(ws0), B, Lw, st))
    torch.cuda.synchronize()
    d1, d0 = ws1.double().sum(0)[:79], ws0.double().sum(0)[:79]
    assert float((d1 - d0).abs().max()) < 1e-2 * float(d0.abs().max()), float((d1 - d0).abs().max()) / float(d0.abs().max())

    # g rebuilt on load from the pooled gradient (MaxPool1d(4) + ReLU backward): against sed_maxpool4_relu_bwd -> g -> the kernel above
    if L1 >= 4:
        Ho = L1 // 4
        dyp = torch.randn(B // 8, Ho, 8, 64, device=dev, generator=g).to(bf)
        sc, sh = torch.rand(64, device=dev, generator=g) + 0.5, torch.randn(64, device=dev, generator=g) * 0.3
        mu, isd = torch.randn(64, device=dev, generator=g) * 0.1, torch.rand(64, device=dev, generator=g) + 0.5
        gfull = torch.full((B // 8, L1, 8, 64), 7.0, device=dev, dtype=bf)
        npp = lib.sed_maxpool4_bwd_nparts(B // 8, L1, 8, 64)
        pa, pb = torch.zeros(npp, 2, 64, device=dev), torch.zeros(npp, 2, 64, device=dev)
        L.check(lib.sed_maxpool4_relu_bwd(1, P(dyp), P(z0), P(sc), P(sh), P(mu), P(isd), P(gfull), P(pa), B // 8, L1, 8, 64, st))
        L.check(lib.sed_maxpool4_relu_bwd(1, P(dyp), P(z0), P(sc), P(sh), P(mu), P(isd), None, P(pb), B // 8, L1, 8, 64, st))
        ws2 = torch.full((npt, 80, 64), 5.0, device=dev)
        L.check(lib.sed_m5_conv1_wgrad_fused(1, P(x), P(gfull), P(z0), P(ca), P(cb), P(cc), P(ws2), B, Lw, st))
        ws3 = torch.full((npt, 80, 64), 5.0, device=dev)
        L.check(lib.sed_m5_conv1_wgrad_fused_pool(1, P(x), P(dyp), P(z0), P(sc), P(sh), P(ca), P(cb), P(cc), P(ws3), B, Lw, st))
        torch.cuda.synchronize()
        assert torch.equal(pa, pb)                                     # the statistics do not depend on whether g is stored
        d2, d3 = ws2.double().sum(0)[:79], ws3.double().sum(0)[:79]
        assert float((d3 - d2).abs().max()) < 1e-5 * float(d2.abs().max()) + 1e-6        # same bf16 dz, same MFMAs


@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 45, 64, 64), (3, 33, 64, 128), (2, 70, 128, 128), (1, 30, 256, 256)])
def test_column_taps_variant_is_bit_identical(L, B, H, Cin, Cout):
    """sed_conv3x3_fwd_col (taps 1, 4, 7 only; W = 8 interleaved Conv1d layout of M5) against sed_conv3x3_fwd on 3x3 weights
    whose side columns are zero: the skipped products are exact zeros, so the outputs are the same bits and the statistics the same sums."""
    lib, P, dev, bf = L.lib(), L.ptr, "cuda", torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    W = 8
    g = torch.Generator(device="cuda").manual_seed(B * 17 + H)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g).to(bf)
    zref = torch.randn(B, H, W, Cout, device=dev, generator=g).to(bf)
    w = torch.zeros(Cout, Cin, 3, 3, device=dev)
    w[:, :, :, 1] = torch.randn(Cout, Cin, 3, device=dev, generator=g) * 0.05
    sc_i, sh_i = torch.rand(Cin, device=dev, generator=g) + 0.5, torch.randn(Cin, device=dev, generator=g) * 0.3
    sc_o, sh_o = torch.rand(Cout, device=dev, generator=g) + 0.5, torch.randn(Cout, device=dev, generator=g) * 0.3
    mean, invstd = torch.randn(Cout, device=dev, generator=g) * 0.1, torch.rand(Cout, device=dev, generator=g) + 0.5
    wpack = torch.empty(9 * Cin * Cout, device=dev, dtype=bf)
    L.check(lib.sed_pack_conv_weight(1, P(w), P(wpack), Cout, Cin, Cout, Cin, 0, st))
    nparts = lib.sed_conv_nparts(B, H, W)
    for pro, epi in ((0, 0), (1, 1), (0, 1), (1, 0), (0, 2)):
        outs = []
        for fn in (lib.sed_conv3x3_fwd, lib.sed_conv3x3_fwd_col):
            out = torch.full((B, H, W, Cout), 7.0, device=dev, dtype=bf)
            part = torch.full((nparts, 2, Cout), 3.0, device=dev)
            L.check(fn(1, pro, epi, P(x), P(sc_i) if pro else None, P(sh_i) if pro else None, P(wpack), P(out),
                       P(zref) if epi == 2 else None, P(sc_o) if epi == 2 else None, P(sh_o) if epi == 2 else None,
                       P(mean) if epi == 2 else None, P(invstd) if epi == 2 else None, P(part) if epi else None, B, H, W, Cin, Cout, st))
            torch.cuda.synchronize()
            outs.append((out, part))
        assert torch.equal(outs[0][0], outs[1][0]), (pro, epi)
        if epi:
            # (the two entry points may cut the image into different strips -- 128-channel layers take all output channels in one
            #  workgroup on the 3x3 path: the per-strip rows differ, their fixed-order column sums agree to fp32 rounding)
            a, b = outs[0][1].double().sum(0), outs[1][1].double().sum(0)
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), (pro, epi)


@pytest.mark.parametrize("B,H,W,Cd,C", [(2, 37, 32, 64, 32), (3, 50, 16, 128, 64), (2, 45, 8, 128, 128), (1, 9, 16, 64, 64)])
def test_pool_backward_statistics_from_pooled_tensors(L, B, H, W, Cd, C):
    """sed_conv3x3_dgrad_poolstats: the data gradient that produces a pooled block output's gradient dy also yields the pool +
    ReLU + BatchNorm backward statistics (sum g, sum g*xhat) of that block from dy, the pooled activation and the
    active-pixel counts of the forward -- against sed_pool_relu_bwd_stats on the full-resolution z (same dy), and the
    device-side fallback for a channel with gamma = 0.  Geometry: dy [B][H][W][C], z [B][2H+1][2W][C] (odd height: the
    pooling floor drops the last row)."""
    lib, P, dev, bf = L.lib(), L.ptr, "cuda", torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(7 * B + H)
    Hz, Wz = 2 * H + 1, 2 * W
    z = torch.randn(B, Hz, Wz, C, device=dev, generator=g).to(bf)
    gamma = torch.rand(C, device=dev, generator=g) + 0.5
    gamma[1] = -gamma[1]
    beta = torch.randn(C, device=dev, generator=g) * 0.3
    mean, invstd = torch.randn(C, device=dev, generator=g) * 0.1, torch.rand(C, device=dev, generator=g) + 0.5
    scale = gamma * invstd
    shift = beta - mean * scale
    dz = torch.randn(B, H, W, Cd, device=dev, generator=g).to(bf)
    w = torch.randn(Cd, C, 3, 3, device=dev, generator=g) * 0.05          # conv weight [out = Cd][in = C]: its data gradient maps Cd -> C
    wpack_t = torch.empty(9 * Cd * C, device=dev, dtype=bf)
    L.check(lib.sed_pack_conv_weight(1, P(w), P(wpack_t), Cd, C, Cd, C, 1, st))
    assert lib.sed_dgrad_poolstats_supported(1, W, Cd, C)

    def run(scale_, shift_):
        y = torch.empty(B, H, W, C, device=dev, dtype=bf)
        cnt = torch.empty(B, H, W, C, device=dev, dtype=torch.uint8)
        L.check(lib.sed_bn_relu_pool_cnt_fwd(1, P(z), P(scale_), P(shift_), P(y), P(cnt), B, Hz, Wz, C, st))
        y_ref = torch.empty_like(y)
        L.check(lib.sed_bn_relu_pool_fwd(1, P(z), P(scale_), P(shift_), P(y_ref), B, Hz, Wz, C, 2, st))
        assert torch.equal(y, y_ref)
        act = (z[:, :2 * H].float() * scale_ + shift_ > 0).view(B, H, 2, W, 2, C).sum(dim=(2, 4))
        assert torch.equal(cnt.long(), act)
        n_old = lib.sed_pool_bwd_nparts(B, Hz, Wz, C)
        nparts = lib.sed_conv_nparts(B, H, W)
        dy = torch.full((B, H, W, C), 7.0, device=dev, dtype=bf)
        part = torch.full((nparts, 2, C), 3.0, device=dev)
        flag = torch.zeros(1, device=dev, dtype=torch.int32)
        L.check(lib.sed_conv3x3_dgrad_poolstats(1, P(dz), P(wpack_t), P(dy), P(y), P(cnt), P(scale_), P(shift_), P(mean), P(invstd),
                                                P(part), nparts, P(flag), B, H, W, Cd, C, st))
        dy_ref = torch.empty_like(dy)
        L.check(lib.sed_conv3x3_fwd(1, 0, 0, P(dz), None, None, P(wpack_t), P(dy_ref), None, None, None, None, None, None, B, H, W, Cd, C, st))
        assert torch.equal(dy, dy_ref)
        old = torch.full((n_old, 2, C), 5.0, device=dev)
        L.check(lib.sed_pool_relu_bwd_stats(1, P(dy), P(z), P(scale_), P(shift_), P(mean), P(invstd), P(old), B, Hz, Wz, C, 2, st))
        return part, flag, old.sum(0), dy, y, nparts

    part, flag, old, dy, y, nparts = run(scale, shift)
    assert int(flag.item()) == 0
    new = part.sum(0)
    mag = (dy.float().abs().view(-1, C).sum(0) * 0.25)                     # sum |g| bound per channel
    torch.testing.assert_close(new[0], old[0], rtol=1e-3, atol=1e-3)
    # sum g*xhat: the pooled activation is bf16 (2^-9 relative, random sign); xhat = O(1)
    assert ((new[1] - old[1]).abs() <= 2e-3 * mag * 4 + 2e-2 * old[1].abs()).all(), (new[1] - old[1]).abs().max()

    # gamma = 0 in one channel: flag raised, the conditional per-pixel pass overwrites every partial row
    scale0 = scale.clone()
    scale0[3] = 0.0
    beta0 = beta.clone()
    beta0[3] = 0.4                      # (with beta <= 0 the channel is dead: g = 0, nothing to recover, no flag)
    shift0 = beta0 - mean * scale0
    part, flag, old, dy, y, nparts = run(scale0, shift0)
    assert int(flag.item()) == 1
    L.check(lib.sed_pool_relu_bwd_stats_if(P(flag), 1, P(dy), P(z), P(scale0), P(shift0), P(mean), P(invstd), P(part), nparts,
                                           B, Hz, Wz, C, 2, st))
    # (same per-pixel arithmetic; the conditional pass runs with fewer workgroups, so its fp32 partial sums group differently)
    torch.testing.assert_close(part.sum(0), old, rtol=1e-4, atol=1e-4 * float(old.abs().max()))
    # flag = 0: the conditional launch leaves the partials alone
    keep = part.clone()
    flag.zero_()
    part.fill_(1.25)
    L.check(lib.sed_pool_relu_bwd_stats_if(P(flag), 1, P(dy), P(z), P(scale0), P(shift0), P(mean), P(invstd), P(part), nparts,
                                           B, Hz, Wz, C, 2, st))
    assert torch.equal(part, torch.full_like(keep, 1.25))

    # |beta| >> |gamma| (ratio 50): y is about beta*cnt/4, the subtraction of the pooled form cancels and amplifies the bf16 rounding
    # of y by beta/gamma -- the kernel must hand the layer to the per-pixel pass (flag), whose sums then equal the reference pass;
    # a ratio of 4 stays on the pooled form and inside its tolerance
    for ratio, want_flag in ((50.0, 1), (4.0, 0)):
        gamma2 = gamma.clone()
        beta2 = beta.clone()
        gamma2[3] = 0.01
        beta2[3] = 0.01 * ratio
        scale2 = gamma2 * invstd
        shift2 = beta2 - mean * scale2
        part, flag, old, dy, y, nparts = run(scale2, shift2)
        assert int(flag.item()) == want_flag, (ratio, int(flag.item()))
        if want_flag:
            L.check(lib.sed_pool_relu_bwd_stats_if(P(flag), 1, P(dy), P(z), P(scale2), P(shift2), P(mean), P(invstd), P(part), nparts,
                                                   B, Hz, Wz, C, 2, st))
            torch.testing.assert_close(part.sum(0), old, rtol=1e-4, atol=1e-4 * float(old.abs().max()))
        else:
            new = part.sum(0)
            mag = (dy.float().abs().view(-1, C).sum(0) * 0.25)
            assert ((new[1] - old[1]).abs() <= 2e-3 * mag * 4 * max(1.0, ratio) + 2e-2 * old[1].abs()).all()
