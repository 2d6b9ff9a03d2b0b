"""CPU oracle of the CRNN variant (BASELINE.json configs[3]):

    conv_blocks -> mean(dim=3) -> transpose -> nn.GRU(C_last, 256, batch_first, bidirectional)
                -> Linear(512, classes) -> interpolate(2**num_pools)

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

The reference repository has NO recurrent model: the specification above is the
survey's, and the arithmetic is torch.nn.GRU's.  PARITY UNPINNED BY THE REFERENCE; pinned instead
against torch.nn.GRU itself: `gru_bidir_fwd` / `gru_bidir_bwd` below restate the published GRU
equations explicitly (gate order r, z, n; h' = (1-z) n + z h) and tests/test_crnn_oracle.py checks
them against torch.nn.GRU forward and autograd on CPU.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from . import cnn_oracle as CO

GRU_SUFFIXES = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
                "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse")


def make_state(classes_num, model_config, hidden=256, seed=0) -> Dict[str, torch.Tensor]:
    """CNN part: the reference's init (cnn_oracle.make_state).  GRU: torch defaults U(+-1/sqrt(hidden));
    FC (classes, 2*hidden): the reference's init_layer (kaiming-uniform, zero bias)."""
    sd = CO.make_state(classes_num, model_config, seed=seed)
    g = torch.Generator().manual_seed(seed + 7919)
    c_last = model_config[-1][0]
    k = 1.0 / hidden ** 0.5
    for sfx in GRU_SUFFIXES:
        if sfx.startswith("weight_ih"):
            shape = (3 * hidden, c_last)
        elif sfx.startswith("weight_hh"):
            shape = (3 * hidden, hidden)
        else:
            shape = (3 * hidden,)
        sd["gru." + sfx] = (torch.rand(shape, generator=g) * 2 - 1) * k
    w = torch.empty(classes_num, 2 * hidden)
    CO.kaiming_uniform_leaky_relu_(w, g)
    sd["event_fc.weight"] = w
    sd["event_fc.bias"] = torch.zeros(classes_num)
    return sd


def param_names(n_blocks):
    base = [n for n in CO.param_names(n_blocks) if not n.startswith("event_fc")]
    return base + ["gru." + s for s in GRU_SUFFIXES] + ["event_fc.weight", "event_fc.bias"]


# ---- explicit GRU (what the HIP kernels implement) ------------------------------------------------
def gru_dir_fwd(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """x (B, t, In) -> h (B, t, H) and the per-step cache (r, z, n, ghn, h_prev)."""
    B, t, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.t() + b_ih                       # all steps at once
    h = x.new_zeros(B, H)
    hs = x.new_zeros(B, t, H)
    cache = []
    order = range(t - 1, -1, -1) if reverse else range(t)
    for tt in order:
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, tt, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, tt, H:2 * H] + gh[:, H:2 * H])
        ghn = gh[:, 2 * H:]
        n = torch.tanh(gi[:, tt, 2 * H:] + r * ghn)
        h_prev = h
        h = (1 - z) * n + z * h_prev
        hs[:, tt] = h
        cache.append((tt, r, z, n, ghn, h_prev))
    return hs, cache


def gru_dir_bwd(dhs, x, w_ih, w_hh, cache):
    """Back-propagation through time for one direction.  Returns dx, dW_ih, dW_hh, db_ih, db_hh."""
    B, t, H = dhs.shape
    dgi = dhs.new_zeros(B, t, 3 * H)
    dgh = dhs.new_zeros(B, t, 3 * H)
    hprev = dhs.new_zeros(B, t, H)
    dh_carry = dhs.new_zeros(B, H)
    for (tt, r, z, n, ghn, h_prev) in reversed(cache):
        dh = dhs[:, tt] + dh_carry
        dn_pre = dh * (1 - z) * (1 - n * n)
        dz_pre = dh * (h_prev - n) * z * (1 - z)
        dr_pre = dn_pre * ghn * r * (1 - r)
        dgi[:, tt] = torch.cat([dr_pre, dz_pre, dn_pre], dim=1)
        dgh[:, tt] = torch.cat([dr_pre, dz_pre, dn_pre * r], dim=1)
        hprev[:, tt] = h_prev
        dh_carry = dh * z + dgh[:, tt] @ w_hh
    dgi2, dgh2 = dgi.reshape(B * t, 3 * H), dgh.reshape(B * t, 3 * H)
    dW_ih = dgi2.t() @ x.reshape(B * t, -1)
    dW_hh = dgh2.t() @ hprev.reshape(B * t, H)
    dx = (dgi2 @ w_ih).reshape(B, t, -1)
    return dx, dW_ih, dW_hh, dgi2.sum(0), dgh2.sum(0)


def gru_bidir_fwd(x, sd):
    hf, cf = gru_dir_fwd(x, sd["gru.weight_ih_l0"], sd["gru.weight_hh_l0"], sd["gru.bias_ih_l0"],
                         sd["gru.bias_hh_l0"], False)
    hr, cr = gru_dir_fwd(x, sd["gru.weight_ih_l0_reverse"], sd["gru.weight_hh_l0_reverse"],
                         sd["gru.bias_ih_l0_reverse"], sd["gru.bias_hh_l0_reverse"], True)
    return torch.cat([hf, hr], dim=2), (cf, cr)


def gru_bidir_bwd(dh, x, sd, caches):
    H = dh.shape[2] // 2
    grads = {}
    dx = 0
    for d, sfx in enumerate(("", "_reverse")):
        dxi, dwi, dwh, dbi, dbh = gru_dir_bwd(dh[:, :, d * H:(d + 1) * H], x, sd["gru.weight_ih_l0" + sfx],
                                              sd["gru.weight_hh_l0" + sfx], caches[d])
        dx = dx + dxi
        grads["gru.weight_ih_l0" + sfx], grads["gru.weight_hh_l0" + sfx] = dwi, dwh
        grads["gru.bias_ih_l0" + sfx], grads["gru.bias_hh_l0" + sfx] = dbi, dbh
    return dx, grads


# ---- whole-model autograd stepper (ATen kernels + torch.nn.functional GRU via nn.GRU) ---------------
class CrnnAutogradStepper:
    def __init__(self, sd, model_config, recall_factor, lr, hidden=256):
        self.cfg = list(model_config)
        self.names = param_names(len(self.cfg))
        self.params = {k: sd[k].clone().requires_grad_(True) for k in self.names}
        self.buffers = {k: v.clone() for k, v in sd.items() if k not in self.params}
        self.pos_weight = torch.tensor([float(recall_factor)])
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr, betas=(0.9, 0.999), eps=1e-8,
                                    weight_decay=0.0, amsgrad=True)
        self.ratio = 2 ** CO.num_pools_of(self.cfg)
        self.hidden = hidden
        self.iterations = 0

    def features(self, x, training=True):
        a = x
        for i, (_, pool) in enumerate(self.cfg):
            for j in (1, 2):
                pre = f"conv_blocks.{i}"
                a = F.conv2d(a, self.params[f"{pre}.conv{j}.weight"], None, 1, 1)
                a = F.batch_norm(a, self.buffers[f"{pre}.bn{j}.running_mean"], self.buffers[f"{pre}.bn{j}.running_var"],
                                 self.params[f"{pre}.bn{j}.weight"], self.params[f"{pre}.bn{j}.bias"], training,
                                 CO.BN_MOMENTUM, CO.BN_EPS)
                a = F.relu_(a)
            a = CO.avgpool_fwd(a, pool)
        return a.mean(dim=3).transpose(1, 2)

    def forward(self, x, training=True):
        m = self.features(x, training)
        flat = [self.params["gru." + s] for s in GRU_SUFFIXES]
        h0 = m.new_zeros(2, m.shape[0], self.hidden)
        out, _ = torch._VF.gru(m, h0, flat, True, 1, 0.0, False, True, True)
        pre_logits = F.linear(out, self.params["event_fc.weight"], self.params["event_fc.bias"])
        return CO.interpolate(pre_logits, self.ratio)

    def step(self, x, y):
        out = self.forward(x, True)
        N = min(out.shape[1], y.shape[1])
        loss = F.binary_cross_entropy_with_logits(out[:, :N], y[:, :N], pos_weight=self.pos_weight)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        self.iterations += 1
        if self.iterations % 200 == 0:
            for g in self.opt.param_groups:
                g["lr"] *= 0.997
        return loss.detach()
