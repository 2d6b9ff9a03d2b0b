"""Parity at the BASELINE.json geometries (the kernels bench.py actually times), plus the product ConvBlock
against the reference's G1 fixture.  Needs the MI355X:  pytest -m gpu.

* config 2 (Cnn_9 bf16, 60 s clips, T = 6001): bf16 / C1-mode train step at B = 2 against
  oracle.cnn_oracle.train_step_grads (fp32 CPU restatement of spectogram_models.py:128-202, common.py:16-30), and
  the size-independent properties at the full B = 32;
* config 4 (CRNN, T = 6001 -> 750 recurrence steps): bf16 loss trajectory + logits against oracle/crnn_oracle.py;
* config 5 (M5, 31680-sample frames): a 64-frame batch against oracle/m5_oracle.py and properties at 2880 frames;
* G1: `ConvBlock(...)` forward + autograd backward (incl. the input gradient) in fp32.

bf16 tolerances: relative L2 on logits (the fp32 1e-3 gate is judged in fp32 mode elsewhere), gradient direction
(cosine) and norm ratio -- element-wise agreement between a bf16 and an fp32 pipeline is not defined (different ReLU
branches wherever a pre-activation lies within bf16 noise of zero)."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cnn_oracle as O
from oracle import cnn_oracle_bf16 as OB

pytestmark = pytest.mark.gpu

MAIN_CFG = [(32, 2), (64, 2), (128, 2), (128, 1)]


@pytest.fixture(scope="module")
def sed():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return importlib.import_module("soundeventdetection-pytorch_amd")


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu().flatten()
    b = torch.as_tensor(b).detach().double().cpu().flatten()
    return float((a - b).norm() / max(b.norm().item(), 1e-30))


def _clip_batch(B, Tn, seed):
    """z-scored log-mel-like features with events (standard normal + marked runs)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 1, Tn, 64, generator=g)
    y = torch.zeros(B, Tn, 1)
    for b in range(B):
        for s in torch.randint(0, Tn - 80, (6,), generator=g).tolist():
            y[b, s:s + 40] = 1.0
            x[b, 0, s:s + 40] += 1.5
    return x, y


# ---------------------------------------------------------------------------------------------
# G1: the product ConvBlock as a standalone autograd node (spectogram_models.py:128-160)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g1_product_convblock_fp32(sed, tag):
    g = load_golden("g1_convblock.npz")
    pool = int(g[f"{tag}.pool"])
    w1 = g[f"{tag}.sd0.conv1.weight"]
    blk = sed.ConvBlock(w1.shape[1], w1.shape[0], pool, precision="fp32")
    sd = {k[len(tag) + 5:]: T(g[k]) for k in g.files if k.startswith(f"{tag}.sd0.")}
    blk.load_state_dict(sd)
    blk.cuda().train()
    x = T(g[f"{tag}.x"]).cuda().requires_grad_()
    y = blk(x)
    assert y.shape == g[f"{tag}.y"].shape
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{tag}.y"], atol=2e-5 * max(1.0, np.abs(g[f"{tag}.y"]).max()), rtol=2e-4)
    y.backward(T(g[f"{tag}.dy"]).cuda())
    ref = g[f"{tag}.dx"]
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref, atol=3e-5 * max(1.0, np.abs(ref).max()), rtol=1e-3)
    for n, p in blk.named_parameters():
        ref = g[f"{tag}.grad.{n}"]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=3e-5 * max(1.0, np.abs(ref).max()), rtol=1e-3, err_msg=n)
    sd1 = blk.state_dict()
    for j in (1, 2):
        for s in ("running_mean", "running_var"):
            np.testing.assert_allclose(sd1[f"bn{j}.{s}"].cpu().numpy(), g[f"{tag}.sd1.bn{j}.{s}"], rtol=2e-4, atol=2e-6)
        assert int(sd1[f"bn{j}.num_batches_tracked"]) == 1
    # eval mode: running statistics, no backward
    blk.eval()
    with torch.no_grad():
        ye = blk(x.detach())
    assert ye.shape == y.shape and torch.isfinite(ye).all()
    with pytest.raises(RuntimeError):
        blk(x.detach().cpu())


# ---------------------------------------------------------------------------------------------
# config 2: bf16, T = 6001, C1 mode, train step vs the oracle
# ---------------------------------------------------------------------------------------------
def test_config2_bf16_T6001_train_step_vs_oracle(sed):
    B, Tn = 2, 6001
    torch.manual_seed(0)
    model = sed.Cnn_AvgPooling(1, MAIN_CFG, precision="bf16")
    with torch.no_grad():       # non-trivial BN affine parameters (init is gamma 1 / beta 0)
        for blk in model.conv_blocks:
            for bn in (blk.bn1, blk.bn2):
                bn.weight.uniform_(0.7, 1.3)
                bn.bias.uniform_(-0.2, 0.2)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x, y = _clip_batch(B, Tn, 77)
    loss_o, logits_o, grads_o, ns_o, _ = O.train_step_grads(x, y, sd, MAIN_CFG, 5.0)
    model.cuda()
    tr = sed.FusedTrainer(model, lr=1e-3, recall_factor=5.0)
    loss = tr.forward_backward(x.cuda(), y.cuda())
    plan = next(iter(model.engine._plans.values()))
    assert plan.c1_mode and plan.c1_dg_fused, "the bench's block-0 dataflow (C1 mode, fused data gradient) must be the one tested"
    logits = model.engine.interpolate(plan)
    assert logits.shape == logits_o.shape == (B, 6000, 1)
    assert rel_l2(logits, logits_o) < 3e-2
    assert abs(loss.item() - float(loss_o)) < 5e-3 * max(1.0, float(loss_o))
    # threshold decisions: identical except where the reference logit is within the bf16 error band of 0
    lg, lo = logits.cpu().numpy(), logits_o.numpy()
    flips = (lg > 0) != (lo > 0)
    assert flips.mean() < 0.02 and (not flips.any() or np.abs(lo[flips]).max() < 0.1)
    # Gradients: against the bf16-STORAGE oracle (oracle/cnn_oracle_bf16.py: the pinned oracle's mathematics in float64,
    # rounded to bf16 exactly where the engine stores bf16, C1-mode formulation of block 0) -- what is left between the two
    # is fp32 summation order and the rare element whose rounding flips, so direction AND size are held tightly.  (Against
    # the plain fp32 oracle only the direction is defined: the two pipelines take different ReLU branches near 0.)
    loss_b, logits_b, grads_b, _ = OB.train_step_grads_bf16(x, y, sd, MAIN_CFG, 5.0)
    assert rel_l2(logits, logits_b) < 4e-3, rel_l2(logits, logits_b)
    assert abs(loss.item() - float(loss_b)) < 5e-4 * max(1.0, float(loss_b))
    worst = {}
    for n in tr.flat.names:
        a, b = tr.flat.G[n].double().cpu().flatten(), grads_b[n].double().flatten()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        worst[n] = (cos, float(a.norm() / b.norm()))
        cos32 = float((a @ grads_o[n].double().flatten()) / (a.norm() * grads_o[n].double().norm() + 1e-30))
        assert cos32 > 0.93, (n, cos32)
    print("config 2 bf16 vs bf16-storage oracle (cos, norm ratio):", {k: (round(c, 5), round(r, 4)) for k, (c, r) in worst.items()})
    for n, (cos, ratio) in worst.items():
        assert cos >= 0.999, (n, cos)
        assert abs(ratio - 1.0) < 2e-2, (n, ratio)
    sd1 = model.state_dict()
    for k, v in ns_o.items():
        if k.endswith("num_batches_tracked"):
            assert int(sd1[k]) == int(v)
            continue
        ref = v.numpy()
        np.testing.assert_allclose(sd1[k].cpu().numpy(), ref, rtol=3e-2, atol=3e-3 * max(1.0, np.abs(ref).max()), err_msg=k)


def test_f16x3_T6001_train_step_vs_oracle_and_fp32_mode(sed):
    """precision="f16x3" (round 6: fp32 tensors, fp16 hi + lo pieces on the 16-bit matrix pipe, csrc/sed_conv_x3.hip) at the bench's frame
    count: 60 s clips, T = 6001.  Here the per-pixel loss gradients are ~2^-27 in block 0 -- far below fp16's range -- so this is the case
    that exercises the gradient-operand exponents the host picks per layer (engine.CnnEngine._grad_dtype).  Gates: logits within 1e-3 of
    the pinned fp32 oracle with bit-exact decisions (north_star), every parameter gradient at relative L2 <= 2e-3 of the oracle's (the
    fp32-MFMA mode's own distance is printed beside it), and within 1e-3 of the fp32 mode's gradients of the same step."""
    B, Tn = 2, 6001
    torch.manual_seed(0)
    model = sed.Cnn_AvgPooling(1, MAIN_CFG, precision="f16x3")
    with torch.no_grad():
        for blk in model.conv_blocks:
            for bn in (blk.bn1, blk.bn2):
                bn.weight.uniform_(0.7, 1.3)
                bn.bias.uniform_(-0.2, 0.2)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x, y = _clip_batch(B, Tn, 77)
    loss_o, logits_o, grads_o, _, _ = O.train_step_grads(x, y, sd, MAIN_CFG, 5.0)
    out = {}
    for prec in ("f16x3", "fp32"):
        m = sed.Cnn_AvgPooling(1, MAIN_CFG, precision=prec)
        m.load_state_dict(sd)
        m.cuda()
        tr = sed.FusedTrainer(m, lr=1e-3, recall_factor=5.0)
        loss = tr.forward_backward(x.cuda(), y.cuda())
        plan = next(iter(m.engine._plans.values()))
        out[prec] = (m.engine.interpolate(plan).cpu(), float(loss.item()), {n: tr.flat.G[n].double().cpu() for n in tr.flat.names})
        del tr, m
        torch.cuda.empty_cache()
    lg, ls, gr = out["f16x3"]
    assert (lg - logits_o).abs().max().item() < 1e-3
    flips = (lg.numpy() > 0) != (logits_o.numpy() > 0)
    assert not flips.any() or np.abs(logits_o.numpy()[flips]).max() < 2e-6
    assert abs(ls - float(loss_o)) < 1e-5 * max(1.0, float(loss_o))
    worst = {}
    for n in gr:
        ref = grads_o[n].double()
        e_x3 = float((gr[n] - ref).norm() / (ref.norm() + 1e-30))
        e_32 = float((out["fp32"][2][n] - ref).norm() / (ref.norm() + 1e-30))
        e_mm = float((gr[n] - out["fp32"][2][n]).norm() / (out["fp32"][2][n].norm() + 1e-30))
        worst[n] = (e_x3, e_32, e_mm)
    print("f16x3 / fp32-MFMA gradient error against the oracle, and f16x3 against the fp32 mode (relative L2):",
          {k: tuple(float(f"{v:.2e}") for v in t) for k, t in worst.items()})
    for n, (e_x3, e_32, e_mm) in worst.items():
        assert e_x3 < 2e-3 and e_mm < 1e-3, (n, e_x3, e_32, e_mm)


def test_config2_pooled_tensor_statistics_match_the_per_pixel_pass(sed, monkeypatch):
    """The pool / ReLU / BN2 backward statistics accumulated in the next block's data-gradient epilogue from pooled tensors
    (default) against the pass over the full-resolution z2 (SED_POOL_STATS=z): same forward, gradients equal to bf16
    rounding of the pooled activation, at the bench geometry (T = 6001, C1 mode)."""
    B, Tn = 2, 6001
    x, y = _clip_batch(B, Tn, 78)
    grads = {}
    for mode in ("p", "z"):
        monkeypatch.setenv("SED_POOL_STATS", mode)
        torch.manual_seed(0)
        model = sed.Cnn_AvgPooling(1, MAIN_CFG, precision="bf16")
        with torch.no_grad():
            for blk in model.conv_blocks:
                for bn in (blk.bn1, blk.bn2):
                    bn.weight.uniform_(0.7, 1.3)
                    bn.bias.uniform_(-0.2, 0.2)
        model.cuda()
        tr = sed.FusedTrainer(model, lr=1e-3, recall_factor=5.0)
        loss = tr.forward_backward(x.cuda(), y.cuda())
        plan = next(iter(model.engine._plans.values()))
        assert plan.pool_fused == ([True, True, True, False] if mode == "p" else [False] * 4)
        grads[mode] = ({n: tr.flat.G[n].double().cpu().flatten() for n in tr.flat.names}, float(loss.item()))
    monkeypatch.delenv("SED_POOL_STATS")
    assert grads["p"][1] == grads["z"][1]                     # the forward is the same arithmetic
    for n, a in grads["p"][0].items():
        b = grads["z"][0][n]
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.9995, (n, cos)
        assert abs(float(a.norm() / b.norm()) - 1.0) < 2e-2, (n, float(a.norm() / b.norm()))


def test_zero_gamma_channel_takes_the_per_pixel_statistics_inside_a_train_step(sed, monkeypatch):
    """A BatchNorm2 channel with gamma = 0 (and beta > 0: the channel is alive) cannot be handled by the pooled-tensor
    statistics: the data-gradient kernel raises the plan's device flag and the conditional per-pixel pass replaces the
    partials -- the gradients then equal those of the all-per-pixel run (same kernels from that point on)."""
    B, Tn = 2, 512
    x, y = _clip_batch(B, Tn, 80)
    grads, flags = {}, {}
    for mode in ("p", "z"):
        monkeypatch.setenv("SED_POOL_STATS", mode)
        torch.manual_seed(0)
        model = sed.Cnn_AvgPooling(1, MAIN_CFG, precision="bf16")
        with torch.no_grad():
            model.conv_blocks[1].bn2.weight[5] = 0.0
            model.conv_blocks[1].bn2.bias[5] = 0.3
        model.cuda()
        tr = sed.FusedTrainer(model, lr=1e-3, recall_factor=5.0)
        tr.forward_backward(x.cuda(), y.cuda())
        plan = next(iter(model.engine._plans.values()))
        flags[mode] = plan.pool_flag.cpu().tolist()
        grads[mode] = {n: tr.flat.G[n].double().cpu().flatten() for n in tr.flat.names}
    monkeypatch.delenv("SED_POOL_STATS")
    assert flags["p"][1] == 1 and flags["p"][0] == 0 and flags["p"][2] == 0 and flags["z"] == [0, 0, 0, 0]
    gz = grads["z"]["conv_blocks.1.bn2.weight"]
    assert abs(float(gz[5])) > 0                                  # dgamma of the gamma = 0 channel is a real number, not 0/0
    for n, a in grads["p"].items():
        b = grads["z"][n]
        assert torch.isfinite(a).all(), n
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.999, (n, cos)
    a5, b5 = float(grads["p"]["conv_blocks.1.bn2.weight"][5]), float(gz[5])
    # (blocks 0 and 2 still take the pooled-tensor form in the 'p' run: their fp32 sums differ from the per-pixel ones in the last
    #  bits, the bf16 tensors downstream then round differently: a few 1e-3 of one channel's dgamma)
    assert abs(a5 - b5) <= 4e-3 * abs(b5) + 1e-6, (a5, b5)


def test_default_width_cnn_fused_statistics_coverage(sed, monkeypatch):
    """The reference's default widths (64-128-256-512, main.py's Cnn_9layers): with the weights streamed from L2 into the
    consumers' registers (round 3) block 3's conv1 data gradient (512 -> 256 at W = 8) no longer exceeds the producer/consumer
    kernel's LDS budget, so blocks 0-2 all take the data-gradient epilogue for their pool / ReLU / BN2 backward statistics (the last
    block has no consumer); with SED_PC_WR=0 block 2 falls back to the per-pixel pass.  Gradients agree with the all-per-pixel run."""
    cfg = [(64, 2), (128, 2), (256, 2), (512, 1)]
    B, Tn = 2, 256
    x, y = _clip_batch(B, Tn, 79)
    grads = {}
    for mode in ("p", "z"):
        monkeypatch.setenv("SED_POOL_STATS", mode)
        torch.manual_seed(0)
        model = sed.Cnn_AvgPooling(1, cfg, precision="bf16").cuda()
        tr = sed.FusedTrainer(model, lr=1e-3, recall_factor=5.0)
        loss = tr.forward_backward(x.cuda(), y.cuda())
        assert torch.isfinite(loss).all()
        plan = next(iter(model.engine._plans.values()))
        assert plan.pool_fused == ([True, True, True, False] if mode == "p" else [False] * 4)
        grads[mode] = {n: tr.flat.G[n].double().cpu().flatten() for n in tr.flat.names}
    monkeypatch.delenv("SED_POOL_STATS")
    for n, a in grads["p"].items():
        b = grads["z"][n]
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.999, (n, cos)


def test_config2_full_batch_properties_bf16(sed):
    """B = 32, T = 6001 -- exactly bench.py's per-GPU workload."""
    B, Tn = 32, 6001
    torch.manual_seed(0)
    model = sed.Cnn_AvgPooling(1, MAIN_CFG, precision="bf16").cuda()
    tr = sed.FusedTrainer(model, lr=1e-3, recall_factor=5.0)
    x, y = _clip_batch(B, Tn, 3)
    x, y = x.cuda(), y.cuda()
    P = tr.flat.tensor_dict()
    eng = model.engine
    l1 = tr.forward_backward(x, y).clone()
    g1 = tr.flat.g.clone()
    l2 = tr.forward_backward(x, y).clone()
    assert torch.equal(l1, l2) and torch.equal(g1, tr.flat.g)            # deterministic reductions
    assert torch.isfinite(tr.flat.g).all() and float(tr.flat.g.abs().max()) > 0
    plan = eng.forward(x, P, training=True, update_running_stats=False)
    ref_logits = eng.interpolate(plan).clone()
    assert ref_logits.shape == (B, 6000, 1)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    plan = eng.forward(x[perm].contiguous(), P, training=True, update_running_stats=False)
    assert rel_l2(eng.interpolate(plan), ref_logits[perm]) < 2e-2           # batch-permutation equivariance
    r = ref_logits.view(B, 750, 8)
    assert torch.equal(r, r[:, :, :1].expand_as(r))                         # x8 interpolation structure
    # the B = 2 sub-batch statistics differ, but eval mode (running stats) is per clip: clip i alone == clip i in the batch
    model.eval()
    with torch.no_grad():
        full = model(x)
        one = model(x[5:6].contiguous())
    assert rel_l2(one, full[5:6]) < 1e-2
    model.train()
    first = tr.train_step(x, y).item()
    for _ in range(5):
        last = tr.train_step(x, y).item()
    assert last < first


# ---------------------------------------------------------------------------------------------
# config 4: CRNN at T = 6001 (750 recurrence steps)
# ---------------------------------------------------------------------------------------------
def _grad_report(named_grads, grads_o, cos_min, norm_tol, skip_zero=True):
    """every parameter gradient against the oracle's: (failures, table) -- the whole table is shown when one fails"""
    bad, rows = [], []
    for n, ga in named_grads:
        b = grads_o[n].double().flatten()
        a = ga.double().cpu().flatten()
        if skip_zero and float(b.norm()) < 1e-9 * b.numel() ** 0.5 and float(a.norm()) < 1e-9 * a.numel() ** 0.5:
            rows.append((n, "zero", ""))
            continue
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        ratio = float(a.norm() / (b.norm() + 1e-30))
        rows.append((n, round(cos, 6), round(ratio, 4)))
        if not (cos >= cos_min and abs(ratio - 1.0) <= norm_tol):
            bad.append(n)
    return bad, rows


def test_config4_crnn_bf16_T6001_vs_oracle(sed):
    """CRNN (Cnn_9 + biGRU-256) on 60 s clips, 750 recurrence steps: logits, the 3-step loss trajectory against the fp32 ATen
    stepper, and EVERY parameter gradient (incl. gru.weight_hh_l0*) of one train step against the bf16-storage oracle
    (oracle/crnn_oracle_bf16.py: float64, rounded where csrc/sed_gru.hip and the conv kernels round) at cosine >= 0.999 and
    2 % in norm -- a sign error in one GRU gradient slice cannot pass."""
    from oracle import crnn_oracle as RO
    from oracle import crnn_oracle_bf16 as RB
    B, Tn = 2, 6001
    sd = RO.make_state(1, MAIN_CFG, hidden=256, seed=0)
    x, y = _clip_batch(B, Tn, 21)
    st = RO.CrnnAutogradStepper(sd, MAIN_CFG, 5.0, 1e-3, hidden=256)
    with torch.no_grad():
        logits_o = st.forward(x, True).clone()
    losses_o = [float(st.step(x, y)) for _ in range(3)]
    model = sed.Crnn_AvgPooling(1, MAIN_CFG, precision="bf16")
    model.load_state_dict(sd)
    model.cuda().train()
    with torch.no_grad():
        out = model(x.cuda())
    assert out.shape == logits_o.shape == (B, 6000, 1)
    assert rel_l2(out, logits_o) < 4e-2
    # ---- one train step, gradient by gradient, against the bf16-storage oracle ----------------------------------------------
    model.load_state_dict(sd)
    tr = sed.FusedTrainer(model, lr=1e-3, recall_factor=5.0)
    xs, ys = x.cuda(), y.cuda()
    loss = tr.forward_backward(xs, ys)
    torch.cuda.synchronize()
    plan = next(iter(model.engine._plans.values()))
    loss_b, logits_b, grads_b, _ = RB.train_step_grads_bf16(x, y, sd, MAIN_CFG, 5.0, c1_mode=bool(plan.c1_mode))
    assert abs(float(loss.item()) - float(loss_b)) < 5e-3 * max(1.0, float(loss_b))
    eng_logits = model.engine.interpolate(plan)
    assert rel_l2(eng_logits, logits_b) < 1e-2, rel_l2(eng_logits, logits_b)
    bad, rows = _grad_report([(n, tr.flat.G[n]) for n in grads_b], grads_b, 0.999, 2e-2)
    assert not bad, (bad, rows)
    # ---- trajectory against the fp32 stepper (reload: the forwards above moved the running statistics) -----------------------
    model.load_state_dict(sd)
    tr = sed.FusedTrainer(model, lr=1e-3, recall_factor=5.0)
    losses = [tr.train_step(xs, ys).item() for _ in range(3)]
    np.testing.assert_allclose(losses, losses_o, rtol=2e-2, atol=5e-3)
    assert losses[-1] < losses[0]


def test_config4_crnn_full_batch_determinism(sed):
    """B = 16 (config 4's per-GPU batch), T = 6001: two identical forward+backward passes are bit-identical."""
    B, Tn = 16, 6001
    torch.manual_seed(0)
    model = sed.Crnn_AvgPooling(1, MAIN_CFG, precision="bf16").cuda()
    tr = sed.FusedTrainer(model, lr=1e-3, recall_factor=5.0)
    x, y = _clip_batch(B, Tn, 9)
    x, y = x.cuda(), y.cuda()
    l1 = tr.forward_backward(x, y).clone()
    g1 = tr.flat.g.clone()
    l2 = tr.forward_backward(x, y).clone()
    assert torch.equal(l1, l2) and torch.equal(g1, tr.flat.g)
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    for n in ("gru.weight_hh_l0", "gru.weight_hh_l0_reverse", "conv_blocks.0.conv1.weight"):
        assert float(tr.flat.G[n].abs().max()) > 0, n


# ---------------------------------------------------------------------------------------------
# config 5: M5 raw waveform at the reference frame length
# ---------------------------------------------------------------------------------------------
def test_config5_m5_bf16_64_frames_vs_oracle(sed):
    """M5 at the reference frame length (31680 samples), 64 frames: against the pinned fp32 oracle in direction (different ReLU /
    arg-max branches within bf16 noise) and against the bf16-storage oracle (oracle/m5_oracle_bf16.py) at cosine >= 0.999 / 2 %
    for every parameter gradient."""
    from oracle import m5_oracle as M
    from oracle import m5_oracle_bf16 as MB
    g7 = load_golden("g7_m5.npz")
    sd = {k[4:]: T(g7[k]) for k in g7.files if k.startswith("sd0.")}
    L_ = 31680                   # waveform_configs.py frame size
    gen = torch.Generator().manual_seed(64)
    x = 0.1 * torch.randn(64, 1, L_, generator=gen)
    y = (torch.rand(64, generator=gen) > 0.7).float()
    x[y > 0] += 0.2 * torch.sin(torch.arange(L_) * 0.05)
    loss_o, logits_o, grads_o, _ = M.train_step_grads(x, y, sd, 5.0)
    m = sed.M5(1, precision="bf16")
    m.load_state_dict(sd)
    m.to("cuda:0").train()
    out = m(x.cuda())
    loss = sed.WeightedBCE(5, False)(out, y.cuda())
    loss.backward()
    assert out.shape == logits_o.shape
    assert rel_l2(out, logits_o) < 8e-2
    assert abs(loss.item() - float(loss_o)) < 2e-2 * max(1.0, float(loss_o))
    named = [(n, p.grad) for n, p in m.named_parameters()]
    for n, ga in named:
        b = grads_o[n].double().flatten()
        if float(b.norm()) < 1e-6 * b.numel() ** 0.5:      # Conv1d biases in front of a BatchNorm: true gradient 0
            continue
        a = ga.double().cpu().flatten()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.9, (n, cos)
    loss_b, logits_b, grads_b, _ = MB.train_step_grads_bf16(x, y, sd, 5.0)
    assert rel_l2(out, logits_b) < 1e-2, rel_l2(out, logits_b)
    assert abs(loss.item() - float(loss_b)) < 5e-3 * max(1.0, float(loss_b))
    # Per block: the head and blocks 4-5 hold 0.999 / 2 %.  Going back through nine layers of MaxPool1d(4) + ReLU the two pipelines
    # (fp32 vs float64 accumulation of the same bf16 values) take a different arg-max / ReLU branch at a few near-ties per layer, and
    # each such switch moves a whole gradient path: measured (tools/diag_m5_bf16.py) at 64 frames 0.9998 (block 4), 0.9983-0.9997
    # (block 3), 0.9956-0.9974 (blocks 1-2; the 64-element conv_block2.4.bias 0.989), at 256 frames 0.9999 / 0.9994-0.9999 /
    # 0.9964-0.9992 -- a noise floor that falls with the number of frames, not a formulation error (against the fp32 oracle the same
    # tensors sit at 0.93-0.97).  A sign or indexing error in one gradient slice gives a cosine far below any of these gates.
    gates = {"fc": (0.999, 2e-2), "conv_block5": (0.999, 2e-2), "conv_block4": (0.999, 2e-2), "conv_block3": (0.998, 2e-2),
             "conv_block2": (0.985, 3e-2), "conv_block1": (0.985, 3e-2)}
    bad_all, rows_all = [], []
    for blk, (cmin, ntol) in gates.items():
        bad, rows = _grad_report([(n, g) for n, g in named if n.split(".")[0] == blk], grads_b, cmin, ntol)
        bad_all += bad
        rows_all += rows
    assert len(rows_all) == len(named)
    assert not bad_all, (bad_all, [r for r in rows_all if r[0] in bad_all])

    # ---- round 5: the same comparison with the branch decisions SHARED at near-ties (oracle/m5_oracle_bf16.py, take_decisions): the
    #      engine's ReLU signs and arg-max positions, rebuilt from the tensors it keeps for its backward (z bf16, scale / shift fp32 of
    #      every layer), are taken by the oracle only where its own pre-activation / window maximum is within 2 bf16 ulps of a tie
    #      (MB.tie_tolerance: ulps of the conv output at the channel's rms magnitude -- about |xhat| < 2^-7, 0.6 % of the elements are
    #      eligible; far from a tie the oracle's own decision stands).  Then EVERY block holds cosine >= 0.999 / 2 % (VERDICT round 4,
    #      task 5) with fewer than 0.1 % of the decisions borrowed.  tools/diag_m5_decisions.py (profiles/r05_b_m5_shared_decisions.txt):
    #      with ALL engine decisions taken every gradient sits at cosine >= 0.99995 -- the whole deficit of the un-shared comparison
    #      above is branch noise (about 7000 of 60 M decisions differ, most of them consequences of an upstream flip).
    plan = next(iter(m.engine._plans.values()))
    decisions = []
    for ly in plan.layers:
        N_, H_, _, C_ = ly.z.shape
        z_e = ly.z.float().permute(0, 2, 3, 1).reshape(N_ * 8, C_, H_).cpu()               # frame n*8 + w, NHWC -> (B, C, L)
        pre_e = torch.addcmul(ly.shift.cpu()[None, :, None], z_e, ly.scale.cpu()[None, :, None])     # fp32, as the kernels' fma
        entry = {"mask": pre_e > 0, "idx": None}
        if ly.pool:
            entry["idx"] = torch.nn.functional.max_pool1d(torch.relu(pre_e), 4, 4, return_indices=True)[1]
        decisions.append(entry)
    stats = {}
    loss_s, logits_s, grads_s, _ = MB.train_step_grads_bf16(x, y, sd, 5.0, take_decisions=decisions, decision_stats=stats)
    borrowed = sum(v[0] for v in stats.values())
    total = sum(v[1] for v in stats.values())
    assert borrowed < 1e-3 * total, (borrowed, total, stats)
    assert max(v[0] / v[1] for v in stats.values()) < 3e-3, stats
    assert rel_l2(out, logits_s) < 1e-2 and abs(loss.item() - float(loss_s)) < 5e-3 * max(1.0, float(loss_s))
    bad, rows = _grad_report(named, grads_s, 0.999, 2e-2)
    assert not bad, (bad, [r for r in rows if r[0] in bad], stats)


def test_config5_m5_full_batch_properties(sed):
    """2880 frames = 64 clips of 60 s at 24 kHz / 31680-sample frames with 50 % overlap... (config 5's per-step batch)."""
    Nf, L_ = 2880, 31680
    torch.manual_seed(0)
    m = sed.M5(1, precision="bf16").to("cuda:0").train()
    gen = torch.Generator().manual_seed(5)
    x = (0.1 * torch.randn(Nf, 1, L_, generator=gen)).cuda()
    y = (torch.rand(Nf, generator=gen) > 0.7).float().cuda()
    crit = sed.WeightedBCE(5, False)
    out1 = m(x)
    l1 = crit(out1, y)
    l1.backward()
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
    out2 = m(x)
    l2 = crit(out2, y)
    l2.backward()
    assert torch.equal(out1, out2) and float(l1) == float(l2)
    for n, p in m.named_parameters():
        assert torch.equal(g1[n], p.grad), n
        assert torch.isfinite(p.grad).all(), n
    # eval mode is per frame: a 64-frame slice alone equals the same frames inside the full batch
    m.eval()
    with torch.no_grad():
        full = m(x)
        part = m(x[128:192].contiguous())
    assert rel_l2(part, full[128:192]) < 1e-2
