"""bf16-STORAGE oracle of the raw-waveform M5 train step (BASELINE.json configs[4]).  TEST INFRASTRUCTURE ONLY.

Only `tests/` may import this file; the product path never does.

Same mathematics as oracle/m5_oracle.py -- M5.forward /root/reference/models/waveform_models.py:58-71 over the layers of
:13-56, WeightedBCE(multi_frame=False) /root/reference/utils/common.py:26-30, autograd backward /root/reference/train.py:102 --
in float64, with every tensor the MI355X engine stores in bf16, or feeds to the matrix pipe in bf16, rounded where the engine
rounds it (soundeventdetection-pytorch_amd/m5_engine.py, csrc/sed_m5.hip, csrc/sed_m5_mfma.hip):

  * every convolution runs on bf16 operands (input / activation and weights rounded), fp32 accumulate, and stores its
    BIAS-FREE pre-BatchNorm output z in bf16 (each Conv1d feeds a BatchNorm1d, which removes the bias again: the engine never
    adds it and gives it a zero gradient; running_mean tracks mean(z) + bias as the reference's does);
  * BatchNorm1d batch statistics are those of z AS STORED; scale / shift / mean / invstd are fp32;
  * the activation relu(scale*z + shift) is recomputed on load and rounded to bf16 where it enters the next matrix product;
    block outputs (after MaxPool1d(4), or after the ReLU for conv_block5) are stored in bf16;
  * head: mean over time and Linear in fp32;
  * backward: d(block output) is stored in bf16; MaxPool1d backward sends it to the FIRST arg-max of the recomputed activation,
    ReLU-gated; BatchNorm backward dz = ca*g + cb*z + cc is produced on load and rounded to bf16 as a matrix operand; data
    gradients are stored in bf16 (ReLU-gated with the mask of the layer below where that layer is not a block output); all
    statistics / weight-gradient sums are fp32 (float64 here).

Parity status: derived from the PINNED oracle m5_oracle.py (golden fixture G7); with rounding switched off (`rb = identity`) it
reproduces m5_oracle.train_step_grads in float64 (tests/test_oracle_bf16_storage.py).
"""
from __future__ import annotations

from typing import Callable, Dict

import torch
import torch.nn.functional as F

from . import m5_oracle as M
from .cnn_oracle_bf16 import _identity, round_bf16  # noqa: F401

F64 = torch.float64


def _bn_coeffs(z, gamma, beta):
    """BatchNorm1d training statistics over (batch, time) of z as given; fp32 coefficients like sed_bn_train_finalize"""
    n = z.shape[0] * z.shape[2]
    mean = z.mean(dim=(0, 2))
    var = ((z * z).mean(dim=(0, 2)) - mean * mean).clamp_min(0.0)
    invstd = (1.0 / torch.sqrt(var + M.BN_EPS)).to(torch.float32).to(F64)
    scale = (gamma.to(torch.float32) * invstd.to(torch.float32)).to(F64)
    shift = (beta.to(torch.float32) - mean.to(torch.float32) * scale.to(torch.float32)).to(F64)
    return dict(mean=mean.to(torch.float32).to(F64), invstd=invstd, scale=scale, shift=shift, n=n, var=var)


def _c(v):
    return v[None, :, None]


def _bf16_ulp(v):
    """spacing of the bf16 grid at |v| (8 significant bits): 2^(floor(log2 |v|) - 7); 0 -> the smallest normal spacing"""
    a = v.abs().clamp_min(2.0 ** -126)
    return torch.exp2(torch.floor(torch.log2(a)) - 7.0)


def tie_tolerance(z, scale):
    """What 2 bf16 ulps of the stored conv output are worth in the pre-activation scale*z + shift: 2 |scale| ulp_bf16(max(|z|, rms_c(z))).
    The ulp is taken at the CHANNEL's rms magnitude at least: the two pipelines' z differ by the bf16 roundings of the layer's inputs
    (a few elements one ulp apart), which moves z by ~2^-8 of its typical size whatever the size of the one element near zero."""
    rms = torch.sqrt((z * z).mean(dim=(0, 2), keepdim=True))
    return 2.0 * _c(scale).abs() * torch.maximum(_bf16_ulp(z), _bf16_ulp(rms)) + 1e-12


def train_step_grads_bf16(x, target, sd: Dict[str, torch.Tensor], recall_factor: float, rb: Callable = round_bf16,
                          alg_first: bool = False, take_decisions=None, decision_stats=None):
    """x (B, 1, L) float32, target (B,) or (B, classes).  Returns (loss, logits, grads, new BN running statistics).
    take_decisions (round 5): one entry per layer, dict(mask=bool (B, C, L) -- the ENGINE's ReLU decisions relu'(scale*z + shift) --,
    idx=int64 (B, C, L/4) or None -- the engine's MaxPool1d arg-max positions along L).  Both pipelines compute the same bf16 values
    with fp32 vs float64 accumulation, so at a near-tie (a pre-activation within 2 bf16 ulps of the conv output -- taken at the channel's
    rms magnitude at least, tie_tolerance() -- of zero; a window whose engine-chosen element is that close to the oracle's maximum) either branch is a correct rounding of the reference's
    /root/reference/models/waveform_models.py:59-71 -- but each switched branch moves a whole gradient path and costs cosine.  There,
    and only there, the oracle takes the engine's branch; decision_stats (a dict, filled in) counts per layer how many decisions were
    BORROWED (differed from the oracle's own) out of how many elements.  Far from a tie the oracle's own decision always stands, so a
    wrong engine decision still shows.
    alg_first: conv_block1's weight gradient in the engine's algebraic form (csrc/sed_m5_mfma.hip, round 4): dW1 = ca*G1 + cb*(w1 .
    Gram) + cc*Sp = the contraction of dz = ca*g + cb*z' + cc with z' the UNROUNDED convolution of the bf16 operands and dz itself
    not rounded (the engine never forms it; opt-in SED_M5_ALG=1); False (the default engine path): dz rounded to bf16 from the stored z."""
    if decision_stats is None:
        decision_stats = {}
    P = {k: v.to(F64) for k, v in sd.items()}
    layers = M.layer_list()
    # which layers are block outputs (the engine keeps a stored y / dy for them): the last conv of every block
    is_out = [i == len(layers) - 1 or layers[i + 1][0].split(".")[0] != layers[i][0].split(".")[0] for i in range(len(layers))]
    new_state: Dict[str, torch.Tensor] = {}
    cache = []
    a = rb(x.to(F64))                                   # conv_block1.0's matrix operand (sed_m5_mfma.hip)
    for li, (conv, bn, cin, cout, k, s, p, pool) in enumerate(layers):
        w = P[conv + ".weight"]
        z = rb(F.conv1d(a, rb(w), None, stride=s, padding=p))
        co = _bn_coeffs(z, P[bn + ".weight"], P[bn + ".bias"])
        n = co["n"]
        new_state[bn + ".running_mean"] = (1 - M.BN_MOMENTUM) * P[bn + ".running_mean"] + M.BN_MOMENTUM * (co["mean"] + P[conv + ".bias"])
        new_state[bn + ".running_var"] = (1 - M.BN_MOMENTUM) * P[bn + ".running_var"] + M.BN_MOMENTUM * co["var"] * (n / max(n - 1, 1))
        new_state[bn + ".num_batches_tracked"] = sd[bn + ".num_batches_tracked"] + 1
        pre = z * _c(co["scale"]) + _c(co["shift"])
        mask = pre > 0
        # what 2 bf16 ulps of the stored z are worth in the pre-activation
        tie = tie_tolerance(z, co["scale"])
        if take_decisions is not None:
            em = take_decisions[li]["mask"]
            borrow = (em != mask) & (pre.abs() <= tie)
            decision_stats[conv + ".relu"] = (int(borrow.sum()), mask.numel())
            mask = torch.where(borrow, em, mask)
        act = torch.where(mask, pre, torch.zeros_like(pre))            # = relu(pre) wherever no decision was borrowed
        idx = None
        if pool:
            yp, idx = F.max_pool1d(act, 4, 4, return_indices=True)      # first arg-max of a window
            if take_decisions is not None and take_decisions[li]["idx"] is not None:
                eidx = take_decisions[li]["idx"]
                at_e = act.gather(2, eidx)
                near = (eidx != idx) & (at_e >= yp - tie.gather(2, idx))
                decision_stats[conv + ".argmax"] = (int(near.sum()), idx.numel())
                idx = torch.where(near, eidx, idx)
                yp = act.gather(2, idx)
            out = rb(yp)
        else:
            out = rb(act)
        cache.append(dict(a=a, z=z, co=co, mask=mask, idx=idx, pool=pool, conv=conv, bn=bn, s=s, p=p, shape=act.shape,
                          is_out=is_out[li]))
        a = out
    feat = a
    m = feat.mean(dim=2)
    logits = m @ P["fc.weight"].t() + P["fc.bias"]
    loss, dlogits = M.weighted_bce(logits, target.to(F64), recall_factor)

    grads: Dict[str, torch.Tensor] = {}
    grads["fc.weight"] = dlogits.t() @ m
    grads["fc.bias"] = dlogits.sum(0)
    Lf = feat.shape[2]
    da = rb(((dlogits @ P["fc.weight"]) / Lf)[:, :, None].expand(-1, -1, Lf))        # last.dy (bf16)
    g_next = None                                        # ReLU-gated data gradient handed to a non-output layer
    for li in reversed(range(len(layers))):
        c = cache[li]
        co = c["co"]
        if c["is_out"]:
            if c["pool"]:
                g = torch.zeros(c["shape"], dtype=F64)
                g.scatter_(2, c["idx"], da)
            else:
                g = da
            g = g * c["mask"].to(F64)
        else:
            g = g_next
        gam = P[c["bn"] + ".weight"]
        n = co["n"]
        xhat = (c["z"] - _c(co["mean"])) * _c(co["invstd"])
        dbeta = g.sum(dim=(0, 2))
        dgamma = (g * xhat).sum(dim=(0, 2))
        grads[c["bn"] + ".bias"], grads[c["bn"] + ".weight"] = dbeta, dgamma
        dz = rb(_c(gam * co["invstd"]) * (g - _c(dbeta) / n - xhat * _c(dgamma) / n))
        w = P[c["conv"] + ".weight"]
        grads[c["conv"] + ".bias"] = torch.zeros_like(P[c["conv"] + ".bias"])
        if li == 0 and alg_first:
            is_ = co["invstd"]
            ca_, cb_ = gam * is_, -gam * is_ * is_ * dgamma / n
            cc_ = -gam * is_ * (dbeta / n - co["mean"] * is_ * dgamma / n)
            z_unr = F.conv1d(c["a"], rb(w), None, stride=c["s"], padding=c["p"])
            dz_w = _c(ca_) * g + _c(cb_) * z_unr + _c(cc_)
            grads[c["conv"] + ".weight"] = torch.nn.grad.conv1d_weight(c["a"], w.shape, dz_w, stride=c["s"], padding=c["p"])
        else:
            grads[c["conv"] + ".weight"] = torch.nn.grad.conv1d_weight(c["a"], w.shape, dz, stride=c["s"], padding=c["p"])
        if li > 0:
            d_in = torch.nn.grad.conv1d_input(c["a"].shape, rb(w), dz, stride=c["s"], padding=c["p"])
            below = cache[li - 1]
            if below["is_out"]:
                da = rb(d_in)
            else:
                g_next = rb(d_in * below["mask"].to(F64))
    return loss, logits, grads, new_state
