"""Raw-waveform M5 on the MI355X against the pinned oracle and the reference fixture G7."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import m5_oracle as M

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_m5.npz"))


def _pkg():
    return importlib.import_module("soundeventdetection-pytorch_amd")


def _sd0():
    return {k[4:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("sd0.")}


def _is_conv_bias(n):
    return n.startswith("conv_block") and n.endswith(".bias") and n.split(".")[1] in ("0", "3")


def _model(precision):
    sed = _pkg()
    m = sed.M5(1, precision=precision)
    m.load_state_dict(_sd0())
    return m.to("cuda:0")


def test_seeded_init_matches_reference():
    sed = _pkg()
    torch.manual_seed(0)
    m = sed.M5(1)
    sd, ref = m.state_dict(), _sd0()
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k


def test_fp32_train_step_matches_reference_and_oracle():
    x, y = torch.from_numpy(G["x"]), torch.from_numpy(G["y"])
    loss_o, logits_o, grads_o, new_state = M.train_step_grads(x, y, _sd0(), 5.0)
    sed = _pkg()
    m = _model("fp32").train()
    out = m(x.cuda())
    loss = sed.WeightedBCE(5, False)(out, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    # north_star gate: logits within 1e-3 of the reference's CPU fp32 result
    np.testing.assert_allclose(out.detach().cpu().numpy(), G["logits"], rtol=0, atol=1e-3)
    assert abs(loss.item() - float(G["loss"])) < 1e-4
    assert torch.equal(out.detach().cpu() > 0, torch.from_numpy(G["logits"]) > 0)      # decisions bit-exact
    for n, p in m.named_parameters():
        g = p.grad.cpu()
        if _is_conv_bias(n):
            assert float(g.abs().max()) == 0.0
            continue
        ref = grads_o[n]
        err = float((g - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
        assert err < 2e-3, (n, err)
    sd = m.state_dict()
    for k, v in new_state.items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v)
        else:
            np.testing.assert_allclose(sd[k].cpu().numpy(), v.numpy(), rtol=1e-4, atol=1e-6, err_msg=k)


def test_fp32_adam_trajectory_matches_reference():
    sed = _pkg()
    x, y = torch.from_numpy(G["x"]).cuda(), torch.from_numpy(G["y"]).cuda()
    m = _model("fp32").train()
    opt = sed.FusedAdamAmsgrad(m, lr=1e-3)
    crit = sed.WeightedBCE(5, False)
    for step in range(3):
        opt.zero_grad()
        loss = crit(m(x), y)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(G["loss3"])) < 1e-3
    sd = m.state_dict()
    for n, _ in m.named_parameters():
        if _is_conv_bias(n):
            continue
        got = sd[n].cpu().numpy().reshape(-1)
        if f"sd3.{n}" in G.files:
            np.testing.assert_allclose(got, G[f"sd3.{n}"].reshape(-1), rtol=2e-3, atol=1e-4, err_msg=n)
        else:
            np.testing.assert_allclose(got[:512], G[f"sd3_head.{n}"], rtol=2e-3, atol=1e-4, err_msg=n)
            np.testing.assert_allclose(got[::97], G[f"sd3_stride.{n}"], rtol=2e-3, atol=1e-4, err_msg=n)


def test_eval_forward_full_frame_and_ragged_batch():
    sd = _sd0()
    gen = torch.Generator().manual_seed(3)
    for k in sd:        # non-trivial running statistics
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=gen) * 0.05
        if k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=gen) * 0.5 + 0.05
    x = torch.randn(5, 1, 31680, generator=gen) * 0.1          # 5 frames: padded to 8 inside
    logits_o, _ = M.forward(x, sd, False)
    sed = _pkg()
    m = sed.M5(1, precision="fp32")
    m.load_state_dict(sd)
    m = m.to("cuda:0").eval()
    with torch.no_grad():
        out = m(x.cuda())
    assert out.shape == (5, 1)
    np.testing.assert_allclose(out.cpu().numpy(), logits_o.numpy(), rtol=0, atol=1e-3)


def test_bf16_step_close_to_fp32():
    # full-size frames: the L=2048 fixture leaves only 16 samples per channel in the last BatchNorm, which
    # amplifies bf16 rounding far beyond what the real frame size (31680 -> 30 steps) shows
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(8, 1, 31680, generator=gen) * 0.1
    y = (torch.rand(8, generator=gen) > 0.5).float()
    _, logits_o, grads_o, _ = M.train_step_grads(x, y, _sd0(), 5.0)
    sed = _pkg()
    m = _model("bf16").train()
    out = m(x.cuda())
    loss = sed.WeightedBCE(5, False)(out, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    rel = float((out.detach().cpu() - logits_o).norm() / logits_o.norm())
    assert rel < 0.08, rel
    for n in ("conv_block1.0.weight", "conv_block3.3.weight", "fc.weight"):
        g = dict(m.named_parameters())[n].grad.cpu().reshape(-1)
        cos = float(torch.dot(g, grads_o[n].reshape(-1)) / (g.norm() * grads_o[n].norm() + 1e-30))
        assert cos > 0.9, (n, cos)


def test_training_batch_must_be_multiple_of_8():
    m = _model("fp32").train()
    with pytest.raises(RuntimeError):
        m(torch.zeros(5, 1, 2048, device="cuda"))


def test_device_batches_match_host_getitem():
    wd = importlib.import_module("soundeventdetection-pytorch_amd.dataset.waveform.waveform_dataset")
    items, waves = wd.synthetic_waveform_task(n_files=3, seconds=6.0, seed=1)
    ds = wd.WaveformDataset(items, val_descriptor="val_", waveforms=waves)
    idx = torch.tensor([0, 5, 17, 1000, 4242, 9, 77, 31], device="cuda")
    x, y = ds.device_batch(idx)
    assert x.shape == (8, 1, 31680) and y.shape == (8,)
    for j, i in enumerate(idx.tolist()):
        xh, yh = ds[i]
        np.testing.assert_array_equal(x[j].cpu().numpy(), xh.astype(np.float32))
        assert float(y[j]) == float(yh)


def test_reference_signature_train_loop_with_m5(tmp_path):
    """train()/eval() of train.py:12-131 drive M5 through the fused trainer (bucketed all-reduce hooks, Adam-amsgrad,
    validation on whole recordings with a ragged frame count)."""
    sed = _pkg()
    wd = importlib.import_module("soundeventdetection-pytorch_amd.dataset.waveform.waveform_dataset")
    np.random.seed(0)
    torch.manual_seed(0)
    items, waves = wd.synthetic_waveform_task(n_files=3, seconds=6.0, seed=1)
    ds = wd.WaveformDataset(items, val_descriptor="val_", waveforms=waves)
    loader = wd.WaveformBatchLoader(ds, 16, device="cuda:0")
    model = sed.M5(1, precision="bf16")
    crit = sed.WeightedBCE(5, False)
    trainer = sed.train.train(model, loader, crit, num_steps=12, lr=1e-3, log_freq=6, outputs_dir=str(tmp_path), device="cuda:0")
    assert trainer.step_count == 12
    import json
    recs = [json.loads(l) for l in open(os.path.join(str(tmp_path), "progress.jsonl"))]
    assert len(recs) == 2 and all(np.isfinite(r["train_loss"]) and np.isfinite(r["val_loss"]) for r in recs)
    assert recs[1]["train_loss"] < recs[0]["train_loss"] + 0.2
    ck = torch.load(os.path.join(str(tmp_path), "checkpoints", "iteration_12.pth"), map_location="cpu")
    assert "conv_block5.3.weight" in ck["model"] and int(ck["model"]["conv_block1.1.num_batches_tracked"]) == 12


def test_zfree_first_block_is_bit_identical_to_the_stored_z_path(monkeypatch):
    """SED_M5_ZFREE=1 (csrc/sed_m5_mfma.hip, round 4; opt-in because measured slower): conv_block1's output is never stored, the
    BatchNorm statistics pass, the fused conv + BN + ReLU + MaxPool forward, the pool-backward statistics and the weight gradient
    recompute it from the waveform with the forward's MFMA sequence -- logits, loss, running statistics and the gradients of blocks 2-5
    must equal the default path bit for bit, block 1's own gradients to fp32 rounding (waveform_models.py:15-24 forward and
    backward), train and eval mode."""
    sed = _pkg()
    L_ = 31680
    g = torch.Generator().manual_seed(11)
    x = (0.1 * torch.randn(16, 1, L_, generator=g)).cuda()
    y = (torch.rand(16, generator=g) > 0.6).float().cuda()
    res = {}
    monkeypatch.setenv("SED_M5_ALG", "0")
    monkeypatch.setenv("SED_M5_POOLSTATS", "0")       # (the default path's pooled-tensor statistics differ from the z pass by y's bf16 rounding)
    for mode in ("0", "1"):
        monkeypatch.setenv("SED_M5_ZFREE", mode)
        sed._lib.lib().sed_config_reload()
        torch.manual_seed(3)
        m = sed.M5(1, precision="bf16").to("cuda:0").train()
        out = m(x)
        loss = sed.WeightedBCE(5, False)(out, y)
        loss.backward()
        plan = next(iter(m.engine._plans.values()))
        assert bool(plan.zfree) == (mode == "1")
        grads = {n: p.grad.clone() for n, p in m.named_parameters()}
        m.eval()
        with torch.no_grad():
            ev = m(x).clone()
        res[mode] = (out.detach().clone(), float(loss), grads, ev, {k: v.clone() for k, v in m.state_dict().items()})
    a, b = res["0"], res["1"]
    assert torch.equal(a[0], b[0]) and a[1] == b[1] and torch.equal(a[3], b[3])
    for n in a[2]:
        if n.startswith("conv_block1."):
            # the pool-backward statistics are summed per workgroup in a different grouping (fp32): BN1's gradient sums and, through the
            # coefficients of dz1, conv1's weight gradient agree to rounding, not bit for bit
            d = float((a[2][n] - b[2][n]).abs().max())
            assert d <= 2e-4 * float(a[2][n].abs().max()) + 1e-12, (n, d)
        else:
            assert torch.equal(a[2][n], b[2][n]), n
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), k


def test_algebraic_first_block_backward_matches_the_default(monkeypatch):
    """SED_M5_ALG=1 (csrc/sed_m5_mfma.hip, round 4; opt-in because measured slower): conv_block1's weight gradient as ca*G1 + cb*(w1 .
    Gram) + cc*Sp -- one pass over z for the statistics and G1 = sum g (x) patch, Gram statistics of the input patches, dz never
    formed.  Everything but conv_block1.0.weight is computed by the same kernels' arithmetic (statistics regrouped: fp32 rounding);
    that gradient differs from the default by the bf16 rounding of dz the default applies (waveform_models.py:15-24 backward)."""
    sed = _pkg()
    L_ = 31680
    g = torch.Generator().manual_seed(12)
    x = (0.1 * torch.randn(16, 1, L_, generator=g)).cuda()
    y = (torch.rand(16, generator=g) > 0.6).float().cuda()
    x[y > 0] += 0.2 * torch.sin(torch.arange(L_, device="cuda") * 0.05)
    res = {}
    monkeypatch.setenv("SED_M5_POOLSTATS", "0")
    for mode in ("0", "1"):
        monkeypatch.setenv("SED_M5_ALG", mode)
        sed._lib.lib().sed_config_reload()
        torch.manual_seed(3)
        m = sed.M5(1, precision="bf16").to("cuda:0").train()
        out = m(x)
        loss = sed.WeightedBCE(5, False)(out, y)
        loss.backward()
        plan = next(iter(m.engine._plans.values()))
        assert bool(plan.alg) == (mode == "1")
        res[mode] = (out.detach().clone(), float(loss), {n: p.grad.double().cpu().flatten() for n, p in m.named_parameters()})
    a, b = res["0"], res["1"]
    assert torch.equal(a[0], b[0]) and a[1] == b[1]
    for n in a[2]:
        u, v = a[2][n], b[2][n]
        if float(u.norm()) == 0.0 and float(v.norm()) == 0.0:
            continue
        cos = float((u @ v) / (u.norm() * v.norm() + 1e-30))
        ratio = float(v.norm() / u.norm())
        if n.startswith("conv_block1."):
            assert cos > 0.9995 and abs(ratio - 1) < 5e-3, (n, cos, ratio)
        else:
            assert torch.equal(u, v), n


def test_pooled_tensor_statistics_match_the_z_pass():
    """sed_maxpool4_pooled_stats (round 4): the MaxPool1d(4) / ReLU / BatchNorm-backward sums (sum g, sum g*xhat) of conv_block1 from the pooled
    tensors (y, dy) against sed_maxpool4_relu_bwd's pass over z -- the autograd backward of /root/reference/models/waveform_models.py:18-24;
    an ill-conditioned channel (|beta| > 8 |gamma|) must raise the flag, and sed_maxpool4_relu_bwd_if then reproduces the z pass exactly."""
    L = _pkg()._lib
    lib, P, dev = L.lib(), L.ptr, "cuda"
    st = torch.cuda.current_stream().cuda_stream
    bf = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(11)
    N, H, W, C = 3, 203, 8, 64                       # (H % 4 = 3: tail rows dropped by the pooling floor)
    Ho = H // 4
    z = torch.randn(N, H, W, C, device=dev, generator=g).to(bf)
    dy = torch.randn(N, Ho, W, C, device=dev, generator=g).to(bf)
    gamma = torch.rand(C, device=dev, generator=g) + 0.5
    gamma[3] = -gamma[3]
    beta = torch.randn(C, device=dev, generator=g) * 0.3
    mean, invstd = torch.randn(C, device=dev, generator=g) * 0.1, torch.rand(C, device=dev, generator=g) + 0.5
    nparts = lib.sed_maxpool4_bwd_nparts(N, H, W, C)

    def run(beta_):
        scale = gamma * invstd
        shift = beta_ - mean * scale
        y = torch.empty(N, Ho, W, C, device=dev, dtype=bf)
        L.check(lib.sed_bn_relu_maxpool4_fwd(1, P(z), P(scale), P(shift), P(y), N, H, W, C, st))
        ref = torch.full((nparts, 2, C), 3.0, device=dev)
        L.check(lib.sed_maxpool4_relu_bwd(1, P(dy), P(z), P(scale), P(shift), P(mean), P(invstd), None, P(ref), N, H, W, C, st))
        part = torch.full((nparts, 2, C), 5.0, device=dev)
        flags = torch.tensor([0, 7], device=dev, dtype=torch.int32)
        L.check(lib.sed_maxpool4_pooled_stats(1, P(dy), P(y), P(scale), P(shift), P(mean), P(invstd), P(part), flags.data_ptr(),
                                              flags.data_ptr() + 4, N, H, W, C, st))
        pooled = part.clone()
        L.check(lib.sed_maxpool4_relu_bwd_if(flags.data_ptr(), 1, P(dy), P(z), P(scale), P(shift), P(mean), P(invstd), P(part), N, H, W, C, st))
        torch.cuda.synchronize()
        return ref.double().sum(0), pooled.double().sum(0), part, ref, flags.cpu(), pooled

    r, pl, part, ref, flags, pooled = run(beta)
    assert int(flags[0]) == 0 and int(flags[1]) == 0             # nothing ill-conditioned; the other step's word was cleared
    assert torch.equal(part, pooled)                              # flag 0: the _if launch left the pooled partials alone
    mag = (dy.float().abs().sum(dim=(0, 1, 2)) + 1e-6).double()
    assert ((pl[0] - r[0]).abs() <= 1e-4 * mag).all(), float(((pl[0] - r[0]).abs() / mag).max())
    # sum g*xhat: y is bf16 (2^-9 relative, random sign), amplified by |beta/gamma| <= ~1 here; xhat = O(1)
    assert ((pl[1] - r[1]).abs() <= 4e-3 * mag + 2e-2 * r[1].abs()).all(), float(((pl[1] - r[1]).abs() / mag).max())
    beta_ill = beta.clone()
    beta_ill[5] = 40.0 * gamma[5]
    r, pl, part, ref, flags, pooled = run(beta_ill)
    assert int(flags[0]) == 0                                     # raised by the statistics launch, consumed AND reset by the _if launch
    assert torch.equal(part, ref)                                 # the fallback recomputed every partial row from z

    # replay with FIXED pointers (what a captured graph or a direct ABI user does): one flag word, flag_clear = NULL.  After an
    # ill-conditioned step the next well-conditioned step must not take the z pass any more
    flag1 = torch.zeros(1, device=dev, dtype=torch.int32)
    y = torch.empty(N, Ho, W, C, device=dev, dtype=bf)
    part = torch.empty(nparts, 2, C, device=dev)

    def step(beta_):
        scale = gamma * invstd
        shift = beta_ - mean * scale
        L.check(lib.sed_bn_relu_maxpool4_fwd(1, P(z), P(scale), P(shift), P(y), N, H, W, C, st))
        L.check(lib.sed_maxpool4_pooled_stats(1, P(dy), P(y), P(scale), P(shift), P(mean), P(invstd), P(part), flag1.data_ptr(), None, N, H, W, C, st))
        pooled = part.clone()
        L.check(lib.sed_maxpool4_relu_bwd_if(flag1.data_ptr(), 1, P(dy), P(z), P(scale), P(shift), P(mean), P(invstd), P(part), N, H, W, C, st))
        torch.cuda.synchronize()
        return torch.equal(part, pooled)

    assert step(beta) is True             # pooled statistics stand
    assert step(beta_ill) is False        # the z pass ran
    assert int(flag1.cpu()[0]) == 0
    assert step(beta) is True             # ... and does not run again on the next step
