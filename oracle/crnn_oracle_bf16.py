"""bf16-STORAGE oracle of the CRNN train step (BASELINE.json configs[3]).  TEST INFRASTRUCTURE ONLY.

Only `tests/` may import this file; the product path never does.

Same mathematics as oracle/crnn_oracle.py (ConvBlock stack of /root/reference/models/spectogram_models.py:153-160, :187-193,
torch.nn.GRU's published equations -- the reference itself has no recurrent model, SURVEY D2 --, Linear, x8 interpolate,
WeightedBCE /root/reference/utils/common.py:16-30, autograd backward /root/reference/train.py:102) in float64, with every
operand the MI355X engine feeds to the matrix pipe in bf16, or stores in bf16, rounded where the engine rounds it
(csrc/sed_gru.hip, engine._gru_forward / _gru_backward):

  * the ConvBlock stack: oracle/cnn_oracle_bf16.py (blocks_forward_bf16 / blocks_backward_bf16);
  * m = mean over mel of the bf16 block output, kept in fp32;
  * input projection (gemm_nt_kernel<bf16>): gi = bf16(m) . bf16(W_ih)^T + b_ih, fp32 accumulate;
  * recurrence (gru_seq_fwd_kernel<bf16>): the state h is fp32, its matrix-pipe copy is bf16: gh = bf16(h) . bf16(W_hh)^T + b_hh;
    gate math, the stored sequence and the saved gates are fp32;
  * Linear head (sed_head_fwd, SED_F32) and the loss: fp32;
  * BPTT (gru_seq_bwd_kernel<bf16>): the step's (dr, dz, dn*r) image is bf16 for the carry product
    dh_carry = dh*z + bf16(dgh) . bf16(W_hh); dgi / dgh go to memory in fp32;
  * the GEMM-shaped rest (gemm_nt_kernel<bf16>): dW_ih = bf16(dgi)^T . bf16(m), dW_hh = bf16(dgh)^T . bf16(h_prev),
    dm = bf16(dgi) . bf16(W_ih); bias gradients are fp32 row sums; d(block output) = bf16(dm / W).

Parity status: derived from oracle/crnn_oracle.py (pinned against torch.nn.GRU forward + autograd,
tests/test_crnn_oracle.py); with rounding switched off (`rb = identity`) it reproduces the explicit fp restatement
(cnn_oracle blocks + crnn_oracle.gru_bidir_fwd / gru_bidir_bwd) to float64 precision (tests/test_oracle_bf16_storage.py).
"""
from __future__ import annotations

from typing import Callable, Dict

import torch

from . import cnn_oracle as O
from . import cnn_oracle_bf16 as OB
from . import crnn_oracle as RO

F64 = torch.float64


def _gru_dir_fwd(m, w_ih, w_hh, b_ih, b_hh, reverse, rb):
    B, t, _ = m.shape
    H = w_hh.shape[1]
    gi = rb(m) @ rb(w_ih).t() + b_ih
    whh = rb(w_hh)
    h = m.new_zeros(B, H)
    hs = m.new_zeros(B, t, H)
    cache = []
    for tt in (range(t - 1, -1, -1) if reverse else range(t)):
        gh = rb(h) @ whh.t() + b_hh
        r = torch.sigmoid(gi[:, tt, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, tt, H:2 * H] + gh[:, H:2 * H])
        ghn = gh[:, 2 * H:]
        n = torch.tanh(gi[:, tt, 2 * H:] + r * ghn)
        h_prev = h
        h = (1 - z) * n + z * h_prev
        hs[:, tt] = h
        cache.append((tt, r, z, n, ghn, h_prev))
    return hs, cache


def _gru_dir_bwd(dhs, m, w_ih, w_hh, cache, rb):
    B, t, H = dhs.shape
    dgi = dhs.new_zeros(B, t, 3 * H)
    dgh = dhs.new_zeros(B, t, 3 * H)
    hprev = dhs.new_zeros(B, t, H)
    carry = dhs.new_zeros(B, H)
    whh = rb(w_hh)
    for (tt, r, z, n, ghn, h_prev) in reversed(cache):
        dh = dhs[:, tt] + carry
        dn_pre = dh * (1 - z) * (1 - n * n)
        dz_pre = dh * (h_prev - n) * z * (1 - z)
        dr_pre = dn_pre * ghn * r * (1 - r)
        dgi[:, tt] = torch.cat([dr_pre, dz_pre, dn_pre], dim=1)
        dgh[:, tt] = torch.cat([dr_pre, dz_pre, dn_pre * r], dim=1)
        hprev[:, tt] = h_prev
        carry = dh * z + rb(dgh[:, tt]) @ whh
    dgi2, dgh2 = dgi.reshape(B * t, 3 * H), dgh.reshape(B * t, 3 * H)
    dW_ih = rb(dgi2).t() @ rb(m.reshape(B * t, -1))
    dW_hh = rb(dgh2).t() @ rb(hprev.reshape(B * t, H))
    dm = (rb(dgi2) @ rb(w_ih)).reshape(B, t, -1)
    return dm, dW_ih, dW_hh, dgi2.sum(0), dgh2.sum(0)


def train_step_grads_bf16(x, target, sd: Dict[str, torch.Tensor], model_config, recall_factor: float,
                          rb: Callable = OB.round_bf16, c1_mode: bool = True):
    """One CRNN training step with the engine's rounding points.  x (B, 1, T, F) float32.  Returns (loss, logits, grads, new
    BN running statistics)."""
    P = {k: v.to(F64) for k, v in sd.items()}
    feat, caches, new_state = OB.blocks_forward_bf16(x, P, sd, model_config, rb, c1_mode)
    B, C, t, Wf = feat.shape
    m = feat.mean(dim=3).transpose(1, 2).contiguous()           # (B, t, C)
    hs, gcache = [], []
    for sfx, rev in (("", False), ("_reverse", True)):
        h, c = _gru_dir_fwd(m, P["gru.weight_ih_l0" + sfx], P["gru.weight_hh_l0" + sfx], P["gru.bias_ih_l0" + sfx],
                            P["gru.bias_hh_l0" + sfx], rev, rb)
        hs.append(h)
        gcache.append(c)
    hcat = torch.cat(hs, dim=2)                                   # (B, t, 2H)
    ratio = 2 ** O.num_pools_of(model_config)
    pre = hcat @ P["event_fc.weight"].t() + P["event_fc.bias"]
    logits = O.interpolate(pre, ratio)
    tgt = target.to(F64)
    loss, _ = O.weighted_bce_fwd(logits, tgt, recall_factor)
    dlogits = O.weighted_bce_bwd(logits, tgt, recall_factor)
    K = dlogits.shape[2]
    dpre = dlogits.reshape(B, t, ratio, K).sum(dim=2)
    grads: Dict[str, torch.Tensor] = {}
    grads["event_fc.weight"] = torch.einsum("btk,bth->kh", dpre, hcat)
    grads["event_fc.bias"] = dpre.sum(dim=(0, 1))
    dh = dpre @ P["event_fc.weight"]                              # (B, t, 2H)
    H = dh.shape[2] // 2
    dm = 0
    for d, sfx in enumerate(("", "_reverse")):
        dmi, dwi, dwh, dbi, dbh = _gru_dir_bwd(dh[:, :, d * H:(d + 1) * H], m, P["gru.weight_ih_l0" + sfx],
                                               P["gru.weight_hh_l0" + sfx], gcache[d], rb)
        dm = dm + dmi
        grads["gru.weight_ih_l0" + sfx], grads["gru.weight_hh_l0" + sfx] = dwi, dwh
        grads["gru.bias_ih_l0" + sfx], grads["gru.bias_hh_l0" + sfx] = dbi, dbh
    dfeat = (dm.transpose(1, 2) / Wf)[:, :, :, None].expand(B, C, t, Wf).contiguous()
    grads.update(OB.blocks_backward_bf16(rb(dfeat), P, model_config, caches, rb))
    return loss, logits, grads, new_state


def train_step_grads_fp(x, target, sd, model_config, recall_factor: float):
    """The same step through the explicit fp restatements (cnn_oracle blocks + crnn_oracle.gru_bidir_*), float64: what the
    identity-rounding form of train_step_grads_bf16 must reproduce."""
    P = {k: (v.to(F64) if v.is_floating_point() else v) for k, v in sd.items()}
    a = x.to(F64)
    bc = []
    for i, (_, pool) in enumerate(model_config):
        a, c = O.conv_block_fwd(a, P, f"conv_blocks.{i}", pool, True, None)
        bc.append(c)
    B, C, t, Wf = a.shape
    m = a.mean(dim=3).transpose(1, 2).contiguous()
    hcat, gc = RO.gru_bidir_fwd(m, P)
    ratio = 2 ** O.num_pools_of(model_config)
    pre = hcat @ P["event_fc.weight"].t() + P["event_fc.bias"]
    logits = O.interpolate(pre, ratio)
    tgt = target.to(F64)
    loss, _ = O.weighted_bce_fwd(logits, tgt, recall_factor)
    dlogits = O.weighted_bce_bwd(logits, tgt, recall_factor)
    dpre = dlogits.reshape(B, t, ratio, -1).sum(dim=2)
    grads = {"event_fc.weight": torch.einsum("btk,bth->kh", dpre, hcat), "event_fc.bias": dpre.sum(dim=(0, 1))}
    dm, gg = RO.gru_bidir_bwd(dpre @ P["event_fc.weight"], m, P, gc)
    grads.update(gg)
    d = (dm.transpose(1, 2) / Wf)[:, :, :, None].expand(B, C, t, Wf).contiguous()
    for i in reversed(range(len(model_config))):
        d, g = O.conv_block_bwd(d, bc[i], P, f"conv_blocks.{i}", model_config[i][1], need_dx=i > 0)
        grads.update(g)
    return loss, logits, grads
