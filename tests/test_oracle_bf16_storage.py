"""oracle/cnn_oracle_bf16.py (the bf16-STORAGE restatement the at-size GPU tests compare the bf16 engine with) against the
pinned oracle/cnn_oracle.py: with the rounding switched off it is the same mathematics to float64 precision (both the
generic and the C1-mode formulation of block 0), with the rounding on it stays within bf16 noise of it."""
import pytest
import torch

from oracle import cnn_oracle as O
from oracle import cnn_oracle_bf16 as OB

CFG = [(8, 2), (16, 2), (16, 1)]


def _case(seed=3, B=2, T=20):
    sd = O.make_state(1, CFG, seed=seed)
    g = torch.Generator().manual_seed(seed)
    for k in sd:
        if k.endswith("bn1.weight") or k.endswith("bn2.weight"):
            sd[k] = torch.rand(sd[k].shape, generator=g) * 0.6 + 0.7
        if k.endswith("bn1.bias") or k.endswith("bn2.bias"):
            sd[k] = torch.rand(sd[k].shape, generator=g) * 0.4 - 0.2
    x = torch.randn(B, 1, T, 64, generator=g)
    y = (torch.rand(B, T, 1, generator=g) > 0.7).float()
    return sd, x, y


@pytest.mark.parametrize("c1", [True, False])
def test_identity_rounding_reproduces_the_pinned_oracle(c1):
    sd, x, y = _case()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    loss_o, logits_o, grads_o, ns_o, _ = O.train_step_grads(x.double(), y.double(), sd64, CFG, 5.0)
    loss, logits, grads, ns = OB.train_step_grads_bf16(x, y, sd, CFG, 5.0, rb=OB._identity, c1_mode=c1)
    # (the engine keeps scale / shift / invstd in fp32: that is the only rounding left)
    assert abs(float(loss) - float(loss_o)) < 1e-6
    torch.testing.assert_close(logits, logits_o, rtol=1e-5, atol=1e-5)
    for k, v in grads_o.items():
        torch.testing.assert_close(grads[k], v, rtol=2e-4, atol=2e-5 * float(v.abs().max()), msg=k)
    for k, v in ns_o.items():
        if v.is_floating_point():
            torch.testing.assert_close(ns[k], v, rtol=1e-6, atol=1e-7, msg=k)


def test_bf16_storage_stays_within_bf16_noise_of_the_fp_oracle():
    sd, x, y = _case(seed=5, B=2, T=48)
    loss_o, logits_o, grads_o, _, _ = O.train_step_grads(x, y, sd, CFG, 5.0)
    loss, logits, grads, _ = OB.train_step_grads_bf16(x, y, sd, CFG, 5.0)
    assert abs(float(loss) - float(loss_o)) < 2e-2
    assert float((logits - logits_o.double()).norm() / logits_o.double().norm()) < 5e-2
    for k, v in grads_o.items():
        a, b = grads[k].flatten(), v.double().flatten()
        assert float((a @ b) / (a.norm() * b.norm() + 1e-30)) > 0.9, k
    # the rounding really happens: a bf16-stored tensor differs from its fp64 value
    assert float((OB.round_bf16(x.double()) - x.double()).abs().max()) > 0


# ---------------------------------------------------------------------------------------------------------------------------
# CRNN (oracle/crnn_oracle_bf16.py) and M5 (oracle/m5_oracle_bf16.py): the same two properties
# ---------------------------------------------------------------------------------------------------------------------------
def _crnn_case(seed=4, B=2, T=40, hidden=32):
    from oracle import crnn_oracle as RO
    sd = RO.make_state(1, CFG, hidden=hidden, seed=seed)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 1, T, 64, generator=g)
    y = (torch.rand(B, T, 1, generator=g) > 0.7).float()
    return sd, x, y


def test_crnn_identity_rounding_reproduces_the_explicit_fp_restatement_and_torch_autograd():
    from oracle import crnn_oracle as RO
    from oracle import crnn_oracle_bf16 as RB
    sd, x, y = _crnn_case()
    loss_e, logits_e, grads_e = RB.train_step_grads_fp(x, y, sd, CFG, 5.0)
    for c1 in (True, False):
        loss, logits, grads, _ = RB.train_step_grads_bf16(x, y, sd, CFG, 5.0, rb=RB.OB._identity, c1_mode=c1)
        assert abs(float(loss) - float(loss_e)) < 1e-6
        torch.testing.assert_close(logits, logits_e, rtol=1e-5, atol=1e-5)
        assert set(grads) == set(grads_e)
        for k, v in grads_e.items():
            torch.testing.assert_close(grads[k], v, rtol=5e-4, atol=5e-5 * float(v.abs().max()) + 1e-12, msg=k)
    # ... and the explicit restatement equals ATen autograd through torch's own GRU (the pin of crnn_oracle.py)
    st = RO.CrnnAutogradStepper({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, CFG, 5.0, 1e-3, hidden=32)
    st.pos_weight = st.pos_weight.double()
    out = st.forward(x.double(), True)
    N = min(out.shape[1], y.shape[1])
    lt = torch.nn.functional.binary_cross_entropy_with_logits(out[:, :N], y.double()[:, :N], pos_weight=st.pos_weight)
    lt.backward()
    assert abs(float(lt.detach()) - float(loss_e)) < 1e-9
    for k, p in st.params.items():
        torch.testing.assert_close(p.grad, grads_e[k], rtol=1e-6, atol=1e-9 + 1e-7 * float(grads_e[k].abs().max()), msg=k)


def test_crnn_bf16_storage_stays_within_bf16_noise():
    from oracle import crnn_oracle_bf16 as RB
    sd, x, y = _crnn_case(seed=6, B=2, T=64)
    loss_e, logits_e, grads_e = RB.train_step_grads_fp(x, y, sd, CFG, 5.0)
    loss, logits, grads, _ = RB.train_step_grads_bf16(x, y, sd, CFG, 5.0)
    assert abs(float(loss) - float(loss_e)) < 2e-2
    assert float((logits - logits_e).norm() / logits_e.norm()) < 5e-2
    for k, v in grads_e.items():
        a, b = grads[k].flatten(), v.flatten()
        assert float((a @ b) / (a.norm() * b.norm() + 1e-30)) > 0.9, k
    assert any(float((grads[k] - grads_e[k]).abs().max()) > 0 for k in grads_e if k.startswith("gru."))


def _m5_case(seed=2, B=3, L=2048):
    from conftest import load_golden
    g7 = load_golden("g7_m5.npz")
    sd = {k[4:]: torch.from_numpy(g7[k]) for k in g7.files if k.startswith("sd0.")}
    g = torch.Generator().manual_seed(seed)
    x = 0.1 * torch.randn(B, 1, L, generator=g)
    y = (torch.rand(B, generator=g) > 0.5).float()
    x[y > 0] += 0.2 * torch.sin(torch.arange(L) * 0.05)
    return sd, x, y


def test_m5_identity_rounding_reproduces_the_pinned_oracle():
    from oracle import m5_oracle as M
    from oracle import m5_oracle_bf16 as MB
    sd, x, y = _m5_case()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    loss_o, logits_o, grads_o, ns_o = M.train_step_grads(x.double(), y.double(), sd64, 5.0)
    loss, logits, grads, ns = MB.train_step_grads_bf16(x, y, sd, 5.0, rb=MB._identity)
    assert abs(float(loss) - float(loss_o)) < 1e-6
    torch.testing.assert_close(logits, logits_o, rtol=1e-5, atol=1e-5)
    for k, v in grads_o.items():
        if k.endswith(".bias") and "conv_block" in k and k.split(".")[1] in ("0", "3"):
            assert float(grads[k].abs().max()) == 0.0 and float(v.abs().max()) < 1e-9, k      # Conv1d bias before a BatchNorm
            continue
        torch.testing.assert_close(grads[k], v, rtol=5e-4, atol=5e-5 * float(v.abs().max()), msg=k)
    for k, v in ns_o.items():
        if v.is_floating_point():
            torch.testing.assert_close(ns[k], v, rtol=1e-5, atol=1e-7, msg=k)


def test_m5_bf16_storage_stays_within_bf16_noise():
    from oracle import m5_oracle as M
    from oracle import m5_oracle_bf16 as MB
    sd, x, y = _m5_case(seed=8, B=4, L=4096)
    loss_o, logits_o, grads_o, _ = M.train_step_grads(x, y, sd, 5.0)
    loss, logits, grads, _ = MB.train_step_grads_bf16(x, y, sd, 5.0)
    assert abs(float(loss) - float(loss_o)) < 5e-2
    assert float((logits - logits_o.double()).norm() / logits_o.double().norm()) < 0.15
    ok = 0
    for k, v in grads_o.items():
        b = v.double().flatten()
        if float(b.norm()) < 1e-6 * b.numel() ** 0.5:
            continue
        a = grads[k].flatten()
        ok += float((a @ b) / (a.norm() * b.norm() + 1e-30)) > 0.8
    assert ok >= 0.8 * len([k for k in grads_o if not (k.endswith('.bias') and 'conv_block' in k and k.split('.')[1] in ('0', '3'))])


def test_m5_shared_decisions_only_at_near_ties():
    """oracle/m5_oracle_bf16.py take_decisions (round 5): offered every decision INVERTED (an engine that is wrong everywhere), the oracle
    borrows only the near-ties -- a tiny fraction -- and its gradients stay within the noise of those few branches; offered its own
    decisions it borrows nothing and reproduces itself exactly."""
    from oracle import m5_oracle_bf16 as MB
    sd, x, y = _m5_case(B=4, L=4096)
    base = MB.train_step_grads_bf16(x, y, sd, 5.0)
    # the oracle's own decisions, recovered the way the GPU test rebuilds the engine's: from stored z / scale / shift
    own, wrong = [], []
    a = MB.round_bf16(x.double())
    P = {k: v.double() for k, v in sd.items()}
    import torch.nn.functional as F
    from oracle import m5_oracle as M
    for (conv, bn, cin, cout, k, s, p, pool) in M.layer_list():
        z = MB.round_bf16(F.conv1d(a, MB.round_bf16(P[conv + ".weight"]), None, stride=s, padding=p))
        co = MB._bn_coeffs(z, P[bn + ".weight"], P[bn + ".bias"])
        pre = z * co["scale"][None, :, None] + co["shift"][None, :, None]
        act = torch.relu(pre)
        idx = F.max_pool1d(act, 4, 4, return_indices=True)[1] if pool else None
        own.append({"mask": pre > 0, "idx": idx})
        wrong.append({"mask": ~(pre > 0), "idx": None if idx is None else (idx // 4) * 4 + (idx % 4 + 1) % 4})
        a = MB.round_bf16(F.max_pool1d(act, 4, 4) if pool else act)
    st = {}
    same = MB.train_step_grads_bf16(x, y, sd, 5.0, take_decisions=own, decision_stats=st)
    assert sum(v[0] for v in st.values()) == 0
    assert float(same[0]) == float(base[0])
    for k_ in base[2]:
        assert torch.equal(same[2][k_], base[2][k_]), k_
    st = {}
    inv = MB.train_step_grads_bf16(x, y, sd, 5.0, take_decisions=wrong, decision_stats=st)
    borrowed, total = sum(v[0] for v in st.values()), sum(v[1] for v in st.values())
    relu_b = sum(v[0] for k_, v in st.items() if k_.endswith(".relu"))
    relu_n = sum(v[1] for k_, v in st.items() if k_.endswith(".relu"))
    assert 0 < relu_b < 2e-2 * relu_n, (relu_b, relu_n)               # only the near-ties (|xhat| < ~2^-7: about 1 % of the elements) were taken from the (always wrong) partner
    assert borrowed < 0.1 * total, (borrowed, total)                   # (arg-max: windows whose maximum is the ReLU's 0 tie exactly, harmlessly)
    for k_ in ("fc.weight", "conv_block3.0.weight", "conv_block1.0.weight"):
        a_, b_ = inv[2][k_].flatten(), base[2][k_].flatten()
        assert float((a_ @ b_) / (a_.norm() * b_.norm())) > 0.7, k_           # (4 frames, EVERY near-tie switched the wrong way: still the same gradient, not noise)
