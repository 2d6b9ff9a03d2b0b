"""The explicit GRU restatement in oracle/crnn_oracle.py against torch.nn.GRU (forward and autograd)
on CPU -- this is what pins the CRNN oracle (the reference repository has no recurrent model)."""
import torch

from oracle import crnn_oracle as RO


def test_explicit_gru_matches_torch_gru_forward_and_backward():
    torch.manual_seed(0)
    B, t, In, H = 3, 11, 8, 32
    gru = torch.nn.GRU(In, H, batch_first=True, bidirectional=True).double()
    sd = {"gru." + k: v.detach().clone() for k, v in gru.state_dict().items()}
    x = torch.randn(B, t, In, dtype=torch.float64, requires_grad=True)
    out, _ = gru(x)
    mine, caches = RO.gru_bidir_fwd(x.detach(), sd)
    assert torch.allclose(out, mine, atol=1e-12)
    dh = torch.randn_like(out)
    out.backward(dh)
    dx, grads = RO.gru_bidir_bwd(dh, x.detach(), sd, caches)
    assert torch.allclose(dx, x.grad, atol=1e-11)
    for k, p in gru.named_parameters():
        assert torch.allclose(grads["gru." + k], p.grad, atol=1e-10), k


def test_crnn_stepper_runs_and_uses_all_parameters():
    cfg = [(4, 2), (8, 2), (8, 2), (8, 1)]
    sd = RO.make_state(1, cfg, hidden=32, seed=0)
    st = RO.CrnnAutogradStepper(sd, cfg, 5.0, 1e-3, hidden=32)
    x = torch.randn(2, 1, 30, 64)
    y = (torch.rand(2, 30, 1) < 0.2).float()
    out = st.forward(x, True)
    assert out.shape == (2, 24, 1)
    l0 = float(st.step(x, y))
    for _ in range(5):
        l = float(st.step(x, y))
    assert l < l0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in st.params.values())
    assert set(RO.param_names(4)) == set(st.params)
