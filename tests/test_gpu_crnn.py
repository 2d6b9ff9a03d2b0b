"""CRNN head (bidirectional GRU) on the MI355X vs oracle/crnn_oracle.py (pinned against torch.nn.GRU):
the GEMM / transpose / recurrence kernels one by one through the C ABI, then the whole model
(forward, loss, every parameter gradient, a few Adam steps).

Tolerances: fp32 mode -- GRU outputs 2e-5 abs (exp/tanh ulp noise through t steps), logits 1e-3,
gradients relative L2 2e-3; bf16 mode -- relative L2 / cosine."""
import importlib

import numpy as np
import pytest
import torch

from oracle import crnn_oracle as RO

pytestmark = pytest.mark.gpu
PKG = "soundeventdetection-pytorch_amd"
TINY_CFG = [(4, 2), (8, 2), (8, 2), (8, 1)]
MAIN_CFG = [(32, 2), (64, 2), (128, 2), (128, 1)]


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available()
    return importlib.import_module(PKG), importlib.import_module(PKG + "._lib")


def rel_l2(a, b):
    a = torch.as_tensor(a).double().cpu().flatten()
    b = torch.as_tensor(b).double().cpu().flatten()
    return float((a - b).norm() / max(b.norm().item(), 1e-30))


@pytest.mark.parametrize("dt,M,N,K,ks,tol", [("fp32", 300, 200, 72, 1, 2e-6), ("fp32", 96, 40, 1000, 8, 2e-6),
                                             ("bf16", 513, 129, 128, 1, 6e-3), ("bf16", 64, 256, 4100, 16, 6e-3)])
def test_gemm_nt(env, dt, M, N, K, ks, tol):
    _, L = env
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g) if ks == 1 else None
    ref = A.double() @ Bm.double().t() + (bias.double() if bias is not None else 0)
    C = torch.full((M, N), float("nan"), device="cuda")
    ws = torch.empty(max(1, L.lib().sed_gemm_nt_ws_floats(M, N, ks)), device="cuda")
    dtype = L.SED_F32 if dt == "fp32" else L.SED_BF16
    Ad, Bd = A.cuda(), Bm.cuda()
    L.check(L.lib().sed_gemm_nt(dtype, L.ptr(Ad), K, L.ptr(Bd), K, L.ptr(bias.cuda()) if bias is not None else None,
                                L.ptr(C), N, M, N, K, ks, L.ptr(ws) if ks > 1 else None, None), "gemm_nt")
    torch.cuda.synchronize()
    assert rel_l2(C, ref) < tol
    # strided output / operand views (how the engine writes gi and reads the transposed gradients)
    C2 = torch.zeros((M, 2 * N + 4), device="cuda")
    L.check(L.lib().sed_gemm_nt(dtype, L.ptr(Ad), K, L.ptr(Bd), K, None, C2.data_ptr() + 4 * N, 2 * N + 4, M, N, K, 1,
                                None, None), "gemm_nt")
    assert rel_l2(C2[:, N:2 * N], A.double() @ Bm.double().t()) < tol and float(C2[:, :N].abs().sum()) == 0.0


def _gemm_tn_ref(A, Bm, seq, shift):
    """C[m][n] = sum_k A[k][m] B[k - shift][n], rows shifted inside sequences of `seq` (a row leaving its sequence contributes zero)"""
    K = A.shape[0]
    Bs = torch.zeros_like(Bm)
    v, o = Bm.view(K // seq, seq, -1), Bs.view(K // seq, seq, -1)
    if shift == 0:
        o.copy_(v)
    elif shift == 1:
        o[:, 1:] = v[:, :-1]
    else:
        o[:, :-1] = v[:, 1:]
    return A.double().t() @ Bs.double(), A.double().sum(0)


@pytest.mark.parametrize("dt,tol", [("fp32", 2e-6), ("bf16", 6e-3)])
def test_gemm_tn_and_its_batched_launch(env, dt, tol):
    """sed_gemm_tn (reduction index on the rows, shifted B rows, column sums) against float64, with and without split-K, through strided
    views; sed_gemm_tn_batch (the BPTT tail's four products in one launch) is bit-identical to the four single calls."""
    _, L = env
    lib = L.lib()
    dtype = L.SED_F32 if dt == "fp32" else L.SED_BF16
    g = torch.Generator().manual_seed(5)
    seq, nseq = 13, 24
    K = seq * nseq
    probs = []       # (A view, lda, B view, ldb, M, N, shift, ks)
    Abig = torch.randn(K, 2 * 96 + 8, generator=g).cuda()
    Bbig = torch.randn(K, 2 * 72, generator=g).cuda()
    for i, (M, N, shift, ks) in enumerate([(96, 40, 0, 1), (96, 72, 1, 4), (96, 40, 0, 4), (96, 72, -1, 3)]):
        a_off, b_off = (i // 2) * 96, (i // 2) * 72
        probs.append((Abig[:, a_off:a_off + M], Bbig[:, b_off:b_off + N], M, N, shift, ks))
    singles = []
    for A, Bm, M, N, shift, ks in probs:
        C = torch.full((M, N + 4), float("nan"), device="cuda")
        cs = torch.full((M,), float("nan"), device="cuda")
        ws = torch.empty(max(1, lib.sed_gemm_tn_ws_floats(M, N, ks)), device="cuda")
        L.check(lib.sed_gemm_tn(dtype, A.data_ptr(), Abig.stride(0), Bm.data_ptr(), Bbig.stride(0), L.ptr(C), N + 4, L.ptr(cs), M, N, K, seq, shift,
                                ks, L.ptr(ws) if ks > 1 else None, None), "gemm_tn")
        torch.cuda.synchronize()
        ref, csref = _gemm_tn_ref(A.cpu().contiguous(), Bm.cpu().contiguous(), seq, shift)
        assert rel_l2(C[:, :N], ref) < tol, (M, N, shift, ks)
        assert rel_l2(cs, csref) < 2e-6
        assert bool(torch.isnan(C[:, N:]).all())          # the pad columns of the strided output are untouched
        singles.append((C, cs))
    ds = (L.GemmTnDesc * 4)()
    outs, keep = [], []
    for e, (A, Bm, M, N, shift, ks) in zip(ds, probs):
        C = torch.full((M, N + 4), float("nan"), device="cuda")
        cs = torch.full((M,), float("nan"), device="cuda")
        ws = torch.empty(max(1, lib.sed_gemm_tn_ws_floats(M, N, ks)), device="cuda")
        keep.append(ws)
        e.A, e.B, e.C, e.colsum, e.workspace = A.data_ptr(), Bm.data_ptr(), L.ptr(C), L.ptr(cs), L.ptr(ws) if ks > 1 else None
        e.lda, e.ldb, e.ldc, e.M, e.N, e.K, e.seq, e.shift, e.ksplit = Abig.stride(0), Bbig.stride(0), N + 4, M, N, K, seq, shift, ks
        outs.append((C, cs))
    import ctypes
    L.check(lib.sed_gemm_tn_batch(dtype, ctypes.cast(ds, ctypes.c_void_p), 4, None), "gemm_tn_batch")
    torch.cuda.synchronize()
    for (C1, s1), (C2, s2), (_, _, M, N, _, _) in zip(singles, outs, probs):
        assert torch.equal(C1[:, :N], C2[:, :N]) and torch.equal(s1, s2)
    assert lib.sed_gemm_tn_batch(dtype, ctypes.cast(ds, ctypes.c_void_p), 9, None) != 0      # more than 8 problems: refused


def test_transpose_shift_and_row_sums(env):
    _, L = env
    B, t, C = 3, 7, 37
    R = B * t
    Rp = (R + 3) // 4 * 4
    src = torch.randn(R, C + 5)
    for shift in (0, 1, -1):
        dst = torch.zeros(C, Rp, device="cuda")
        sd = src.cuda()
        L.check(L.lib().sed_transpose_shift(L.ptr(sd), C + 5, L.ptr(dst), Rp, R, C, t, shift, None), "transpose")
        ref = torch.zeros(B, t, C)
        v = src[:, :C].view(B, t, C)
        if shift == 0:
            ref = v.clone()
        elif shift == 1:
            ref[:, 1:] = v[:, :-1]
        else:
            ref[:, :-1] = v[:, 1:]
        assert torch.equal(dst[:, :R].cpu(), ref.view(R, C).t())
    out = torch.empty(C, device="cuda")
    L.check(L.lib().sed_row_sums(L.ptr(dst), Rp, L.ptr(out), C, R, None), "row_sums")
    np.testing.assert_allclose(out.cpu().numpy(), dst[:, :R].sum(1).cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dt,B,t,In,H,rows", [("fp32", 3, 13, 8, 32, None), ("fp32", 40, 9, 16, 256, None), ("bf16", 5, 21, 128, 256, None),
                                              ("bf16", 5, 21, 128, 256, "4"), ("bf16", 11, 17, 128, 256, "8"), ("bf16", 7, 33, 128, 256, "2")])
def test_gru_recurrence_kernels(env, monkeypatch, dt, B, t, In, H, rows):
    """rows: SED_GRU16_ROWS, the batch rows per workgroup of the bf16 / H = 256 recurrence (csrc/sed_gru.hip: 2 = the default two-row form with
    the gate math split over the half-waves, 4 / 8 = the forms it replaced) -- every form against the same float64 restatement of
    torch.nn.GRU's equations (the reference has no recurrent model), odd batches so that a chunk holds rows past the batch."""
    _, L = env
    if rows is not None:
        monkeypatch.setenv("SED_GRU16_ROWS", rows)
    L.lib().sed_config_reload()
    torch.manual_seed(B + t)
    k = 1 / H ** 0.5
    sd = {}
    for sfx in ("", "_reverse"):
        sd["gru.weight_ih_l0" + sfx] = (torch.rand(3 * H, In) * 2 - 1) * k
        sd["gru.weight_hh_l0" + sfx] = (torch.rand(3 * H, H) * 2 - 1) * k
        sd["gru.bias_ih_l0" + sfx] = (torch.rand(3 * H) * 2 - 1) * k
        sd["gru.bias_hh_l0" + sfx] = (torch.rand(3 * H) * 2 - 1) * k
    x = torch.randn(B, t, In)
    dh = torch.randn(B, t, 2 * H)
    sdd = {kk: v.double() for kk, v in sd.items()}
    ref_h, caches = RO.gru_bidir_fwd(x.double(), sdd)
    ref_dx, ref_g = RO.gru_bidir_bwd(dh.double(), x.double(), sdd, caches)
    dtype = L.SED_F32 if dt == "fp32" else L.SED_BF16
    tdt = torch.float32 if dt == "fp32" else torch.bfloat16
    R = B * t
    # input projection on the host side of the test (the GEMM has its own test); recurrence on the device
    gi = torch.cat([x.view(R, In) @ sd["gru.weight_ih_l0" + s].t() + sd["gru.bias_ih_l0" + s] for s in ("", "_reverse")],
                   dim=1).contiguous().cuda()
    bhh = torch.stack([sd["gru.bias_hh_l0"], sd["gru.bias_hh_l0_reverse"]]).contiguous().cuda()
    n = L.lib().sed_gru_pack_elems(H)
    pf, pb = torch.empty(n, dtype=tdt, device="cuda"), torch.empty(n, dtype=tdt, device="cuda")
    wf, wr = sd["gru.weight_hh_l0"].cuda(), sd["gru.weight_hh_l0_reverse"].cuda()
    L.check(L.lib().sed_gru_pack_weights(dtype, L.ptr(wf), L.ptr(wr), L.ptr(pf), L.ptr(pb), H, None), "pack")
    hseq = torch.full((R, 2 * H), float("nan"), device="cuda")
    saved = torch.empty((R, 8 * H), device="cuda")
    L.check(L.lib().sed_gru_seq_fwd(dtype, L.ptr(gi), L.ptr(bhh), L.ptr(pf), L.ptr(hseq), L.ptr(saved), B, t, H, None), "fwd")
    torch.cuda.synchronize()
    got_h = hseq.view(B, t, 2 * H).cpu()
    if dt == "fp32":
        assert (got_h.double() - ref_h).abs().max() < 2e-5
    else:
        assert rel_l2(got_h, ref_h) < 2e-2
    dgi = torch.full((R, 6 * H), float("nan"), device="cuda")
    dgh = torch.full((R, 6 * H), float("nan"), device="cuda")
    dhd = dh.view(R, 2 * H).contiguous().cuda()
    L.check(L.lib().sed_gru_seq_bwd(dtype, L.ptr(dhd), L.ptr(hseq), L.ptr(saved), L.ptr(pb), L.ptr(dgi), L.ptr(dgh), B, t, H,
                                    None), "bwd")
    torch.cuda.synchronize()
    # weight / bias / input gradients follow from dgi, dgh by plain products (done on the host here)
    dgi_c, dgh_c = dgi.cpu().double(), dgh.cpu().double()
    tol = 3e-5 if dt == "fp32" else 4e-2
    dx = 0
    for d, sfx in enumerate(("", "_reverse")):
        gi_d, gh_d = dgi_c[:, d * 3 * H:(d + 1) * 3 * H], dgh_c[:, d * 3 * H:(d + 1) * 3 * H]
        assert rel_l2(gi_d.t() @ x.view(R, In).double(), ref_g["gru.weight_ih_l0" + sfx]) < tol
        assert rel_l2(gi_d.sum(0), ref_g["gru.bias_ih_l0" + sfx]) < tol
        assert rel_l2(gh_d.sum(0), ref_g["gru.bias_hh_l0" + sfx]) < tol
        hprev = torch.zeros(B, t, H, dtype=torch.float64)
        hd = ref_h[:, :, d * H:(d + 1) * H]
        if d == 0:
            hprev[:, 1:] = hd[:, :-1]
        else:
            hprev[:, :-1] = hd[:, 1:]
        assert rel_l2(gh_d.t() @ hprev.view(R, H), ref_g["gru.weight_hh_l0" + sfx]) < tol
        dx = dx + gi_d @ sdd["gru.weight_ih_l0" + sfx]
    assert rel_l2(dx.view(B, t, In), ref_dx) < tol


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
@pytest.mark.parametrize("cfg,B,Tn,H,K", [(TINY_CFG, 3, 30, 32, 1), (TINY_CFG, 2, 61, 64, 3), (MAIN_CFG, 2, 64, 256, 1)])
def test_crnn_model_fp32_matches_oracle(env, cfg, B, Tn, H, K, prec):
    """prec = "f16x3" (round 6): the conv stack through the split-operand kernels (csrc/sed_conv_x3.hip), the recurrent head as in the fp32 mode."""
    sed, _ = env
    ms = importlib.import_module(PKG + ".models.spectogram_models")
    sd = RO.make_state(K, cfg, hidden=H, seed=3)
    model = ms.Crnn_AvgPooling(K, cfg, precision=prec, gru_hidden=H)
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not [m for m in missing.missing_keys if "num_batches" not in m]
    model.cuda().train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 1, Tn, 64, generator=g)
    y = (torch.rand(B, Tn, K, generator=g) < 0.2).float()
    st = RO.CrnnAutogradStepper({k: v.double() for k, v in sd.items()}, cfg, 5.0, 1e-3, hidden=H)
    st.pos_weight = st.pos_weight.double()
    out_t = st.forward(x.double(), True)
    N = min(out_t.shape[1], Tn)
    loss_t = torch.nn.functional.binary_cross_entropy_with_logits(out_t[:, :N], y.double()[:, :N], pos_weight=st.pos_weight)
    loss_t.backward()
    out = model(x.cuda())
    assert out.shape == out_t.shape
    assert (out.cpu().double() - out_t.detach()).abs().max() < 1e-3
    loss = sed.WeightedBCE(5, True)(out, y.cuda())
    assert abs(loss.item() - float(loss_t)) < 1e-4
    loss.backward()
    for n, p in model.named_parameters():
        l2 = rel_l2(p.grad, st.params[n].grad)
        assert l2 < 3e-3, (n, l2)
    # eval mode forward (running statistics, no saved gates)
    model.eval()
    with torch.no_grad():
        e = model(x.cuda())
    assert torch.isfinite(e).all() and e.shape == out.shape


def test_crnn_training_bf16_tracks_fp32_cpu(env):
    sed, _ = env
    ms = importlib.import_module(PKG + ".models.spectogram_models")
    tr = importlib.import_module(PKG + ".train")
    cfg, H = MAIN_CFG, 256
    sd = RO.make_state(1, cfg, hidden=H, seed=0)
    model = ms.Crnn_AvgPooling(1, cfg, precision="bf16", gru_hidden=H)
    model.load_state_dict(sd, strict=False)
    model.cuda()
    trainer = tr.FusedTrainer(model, lr=1e-3, recall_factor=5.0)
    st = RO.CrnnAutogradStepper(sd, cfg, 5.0, 1e-3, hidden=H)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, 1, 128, 64, generator=g)
    y = (torch.rand(4, 128, 1, generator=g) < 0.2).float()
    lg, lo = [], []
    for _ in range(6):
        lg.append(float(trainer.train_step(x.cuda(), y.cuda())))
        lo.append(float(st.step(x, y)))
    assert lg[-1] < lg[0] and abs(lg[0] - lo[0]) < 5e-3
    assert max(abs(a - b) for a, b in zip(lg, lo)) < 0.05
    # determinism of the whole step (fixed-order reductions everywhere, no atomics)
    model2 = ms.Crnn_AvgPooling(1, cfg, precision="bf16", gru_hidden=H)
    model2.load_state_dict(sd, strict=False)
    model2.cuda()
    t2 = tr.FusedTrainer(model2, lr=1e-3, recall_factor=5.0)
    l2 = [float(t2.train_step(x.cuda(), y.cuda())) for _ in range(6)]
    assert l2 == lg
