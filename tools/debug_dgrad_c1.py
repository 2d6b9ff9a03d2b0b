"""Debug aid: sed_conv3x3_dgrad_c1_stats against a torch restatement under structured inputs (which part is off?)."""
import importlib, sys, os
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sed = importlib.import_module("soundeventdetection-pytorch_amd")
L = sed._lib
lib, P, dev, bf = L.lib(), L.ptr, "cuda", torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
W, C = 64, 32


def run(B, H, mask_ones, w_ident, x_const, zs):
    g = torch.Generator(device="cuda").manual_seed(B * 131 + H)
    dz = torch.randn(B, H, W, C, device=dev, generator=g).to(bf)
    w2 = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05
    if w_ident:
        w2 = torch.zeros(C, C, 3, 3, device=dev)
        w2[:, :, 1, 1] = torch.eye(C, device=dev)
    x1 = torch.randn(B, H, W, device=dev, generator=g) * 3.0 + 1.0
    if x_const:
        x1 = torch.ones(B, H, W, device=dev)
    fmean = torch.randn(W, device=dev, generator=g) if zs else None
    fstd = (torch.rand(W, device=dev, generator=g) + 0.5) if zs else None
    mask = torch.randint(0, 65536, (B, H, W, 2), device=dev, generator=g, dtype=torch.int32)
    if mask_ones:
        mask = torch.full_like(mask, 0xFFFF)
    mask = mask.to(torch.int16)
    wpack_t = torch.empty(9 * C * C, device=dev, dtype=bf)
    L.check(lib.sed_pack_conv_weight(1, P(w2), P(wpack_t), C, C, C, C, 1, st))
    part = torch.full((lib.sed_conv_dgrad_c1_nparts(), 10, C), 9.0, device=dev)
    gd = torch.full((B, H, W, C), 7.0, device=dev, dtype=bf)
    L.check(lib.sed_conv3x3_dgrad_c1_stats_g(1, P(dz), P(wpack_t), P(x1), P(fmean), P(fstd), P(mask), P(part), P(gd), B, H, W, st))
    gold = torch.full((B, H, W, C), 7.0, device=dev, dtype=bf)
    part2 = torch.zeros(lib.sed_conv_nparts(B, H, W), 2, C, device=dev)
    L.check(lib.sed_conv3x3_dgrad_c1(1, P(dz), P(wpack_t), P(gold), P(mask), P(part2), B, H, W, C, st))
    torch.cuda.synchronize()
    d = (gd.float() - gold.float()).abs()
    print("  g new vs old: max abs diff", float(d.max()), "frac bad", float((d > 0.05).float().mean()))
    if float(d.max()) > 0.05:
        bad = (d > 0.05).nonzero()[:6].tolist()
        print("  first bad (b,h,w,c):", bad)
        bh = d.amax(dim=(0, 2, 3)); bw = d.amax(dim=(0, 1, 3)); bc = d.amax(dim=(0, 1, 2))
        print("  bad rows", (bh > 0.05).nonzero().flatten().tolist()[:20], "bad cols", (bw > 0.05).nonzero().flatten().tolist()[:70], "bad ch", (bc > 0.05).nonzero().flatten().tolist())
    got = part.double().sum(0).cpu()
    dzc = dz.float().cpu().permute(0, 3, 1, 2)
    gpre = F.conv_transpose2d(dzc, w2.to(bf).float().cpu(), padding=1)
    mk = mask.cpu().to(torch.int32) & 0xFFFF
    c = torch.arange(C)
    half, bit = (c >> 2) & 1, (c & 3) + 4 * (c >> 3)
    on = ((mk[..., half] >> bit) & 1).permute(0, 3, 1, 2).float()
    gg = (gpre * on).to(bf).double()
    xz = x1 if not zs else (x1 - fmean) / fstd
    xz = xz.to(bf).double().cpu()
    xp = F.pad(xz, (1, 1, 1, 1))
    ref = torch.zeros(10, C, dtype=torch.float64)
    for k in range(9):
        ti, tj = divmod(k, 3)
        ref[k] = (gg * xp[:, None, ti:ti + H, tj:tj + W]).sum(dim=(0, 2, 3))
    ref[9] = gg.sum(dim=(0, 2, 3))
    err = (got - ref).abs().amax(dim=1) / ref.abs().max()
    print(f"B={B} H={H} mask1={mask_ones} wI={w_ident} xc={x_const} zs={zs}: rel err per row:", " ".join(f"{e:.1e}" for e in err.tolist()))
    if err.max() > 1e-2:
        print("  got[9][:8]", got[9][:8].tolist())
        print("  ref[9][:8]", ref[9][:8].tolist())
        print("  got[4][:8]", got[4][:8].tolist())
        print("  ref[4][:8]", ref[4][:8].tolist())


for th in ("8", "4"):
    os.environ["SED_DGRAD_TH"] = th
    L.lib().sed_config_reload()
    print("TH", th)
    run(1, 8, True, True, True, False)
    run(1, 8, False, True, True, False)
    run(1, 8, True, False, True, False)
    run(1, 8, True, True, False, False)
    run(1, 8, True, True, False, True)
    run(2, 37, False, False, False, True)
