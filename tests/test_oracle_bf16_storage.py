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
