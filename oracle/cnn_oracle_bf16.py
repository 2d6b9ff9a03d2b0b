"""bf16-STORAGE oracle for the CNN half of the SED training hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/` may import this file; the product path never does.

Same mathematics as oracle/cnn_oracle.py -- ConvBlock.forward (/root/reference/models/spectogram_models.py:153-160),
Cnn_AvgPooling.forward (:185-202), WeightedBCE (/root/reference/utils/common.py:16-30) and their autograd backward
(/root/reference/train.py:102) -- evaluated in float64 with every tensor that the MI355X engine keeps in bf16 rounded to
bf16 at the point where the engine stores it (pre-BN conv outputs z, the recomputed post-BN/ReLU
activations fed to the matrix pipe, pooled block outputs, the backward tensors dz / g / dy, the packed weights).  All sums
(convolutions, statistics) are float64, i.e. "exact" next to the engine's fp32 accumulation.

Why it exists: an fp32 pipeline and a bf16 pipeline take different ReLU branches wherever a pre-activation lies within
bf16 noise of zero, so the plain fp32 oracle only bounds the bf16 engine loosely (gradient cosine 0.93).  This restatement
rounds where the engine rounds; what is left between the two is fp32-vs-fp64 summation and the rare element whose rounding
flips -- gradients agree to cosine >= 0.999 (tests/test_gpu_at_size.py).

Parity status: derived from the PINNED oracle (cnn_oracle.py, checked against the reference's golden vectors); with
rounding switched off (`rb = identity`) it reproduces cnn_oracle.train_step_grads to float64 precision
(tests/test_oracle_bf16_storage.py).
"""
from __future__ import annotations

from typing import Callable, Dict

import torch

from . import cnn_oracle as O

F64 = torch.float64


def round_bf16(t: torch.Tensor) -> torch.Tensor:
    """round-to-nearest-even to bf16, back to float64 (what a bf16 store + load does)"""
    return t.to(torch.float32).to(torch.bfloat16).to(F64)


def _identity(t: torch.Tensor) -> torch.Tensor:
    return t.to(F64)


def _bn_coeffs(z, gamma, beta):
    """training-mode BatchNorm2d as the engine finalizes it: mean / biased variance over (B, H, W) of z AS GIVEN, fp32
    scale = gamma*invstd and shift = beta - mean*scale (csrc/sed_ops.hip: bn_train_finalize_kernel)."""
    n = z.shape[0] * z.shape[2] * z.shape[3]
    mean = z.mean(dim=(0, 2, 3))
    var = (z * z).mean(dim=(0, 2, 3)) - mean * mean
    var = var.clamp_min(0.0)
    invstd = (1.0 / torch.sqrt(var + O.BN_EPS)).to(torch.float32).to(F64)
    scale = (gamma.to(torch.float32) * invstd.to(torch.float32)).to(F64)
    shift = (beta.to(torch.float32) - mean.to(torch.float32) * scale.to(torch.float32)).to(F64)
    return dict(mean=mean.to(torch.float32).to(F64), invstd=invstd, scale=scale, shift=shift, n=n, var=var)


def _c(v):
    return v[None, :, None, None]


def blocks_forward_bf16(x, P, sd, model_config, rb: Callable = round_bf16, c1_mode: bool = True):
    """The ConvBlock stack (spectogram_models.py:153-160, :187-189) in training mode with the engine's storage rounding.
    P: float64 parameters, sd: the state_dict they came from.  Returns (last pooled block output (B, C, t, W), per-block
    caches for blocks_backward_bf16, new BN running statistics)."""
    new_state: Dict[str, torch.Tensor] = {}
    caches = []
    a_in = x.to(F64)
    for i, (_, pool) in enumerate(model_config):
        pre = f"conv_blocks.{i}"
        c: Dict[str, torch.Tensor] = {}
        w1, w2 = P[f"{pre}.conv1.weight"], P[f"{pre}.conv2.weight"]
        g1, b1 = P[f"{pre}.bn1.weight"], P[f"{pre}.bn1.bias"]
        g2, b2 = P[f"{pre}.bn2.weight"], P[f"{pre}.bn2.bias"]
        first_c1 = (i == 0 and c1_mode)
        if first_c1:
            # C1 mode (csrc/conv_common.h): z1 is never stored; BN1's statistics are those of the exact fp32 convolution
            # (Gram statistics of the input patches), the activation is rebuilt on the matrix pipe from bf16 operands with
            # BN1 folded into the weights: a1 = relu(sum bf16(scale*w1) * bf16(x) + shift), shift carried as hi + lo
            z1 = O.conv3x3_fwd(a_in, w1)
            bn1 = _bn_coeffs(z1, g1, b1)
            wf = rb(w1 * _c(bn1["scale"]).reshape(-1, 1, 1, 1))
            sh_hi = rb(bn1["shift"])
            sh_lo = rb(bn1["shift"] - sh_hi)
            pre1 = O.conv3x3_fwd(rb(a_in), wf) + _c(sh_hi + sh_lo)
            a1 = rb(torch.relu(pre1))
            mask1 = pre1 > 0
            c["in1"] = a_in                      # fp32 input
        else:
            z1 = rb(O.conv3x3_fwd(a_in, rb(w1)))
            bn1 = _bn_coeffs(z1, g1, b1)
            pre1 = z1 * _c(bn1["scale"]) + _c(bn1["shift"])
            a1 = rb(torch.relu(pre1))
            mask1 = pre1 > 0
            c["in1"] = a_in
        z2 = rb(O.conv3x3_fwd(a1, rb(w2)))
        bn2 = _bn_coeffs(z2, g2, b2)
        pre2 = z2 * _c(bn2["scale"]) + _c(bn2["shift"])
        mask2 = pre2 > 0
        y = rb(O.avgpool_fwd(torch.relu(pre2), pool))
        for j, bn in ((1, bn1), (2, bn2)):
            n = bn["n"]
            rm, rv = P[f"{pre}.bn{j}.running_mean"], P[f"{pre}.bn{j}.running_var"]
            new_state[f"{pre}.bn{j}.running_mean"] = (1 - O.BN_MOMENTUM) * rm + O.BN_MOMENTUM * bn["mean"]
            new_state[f"{pre}.bn{j}.running_var"] = (1 - O.BN_MOMENTUM) * rv + O.BN_MOMENTUM * bn["var"] * (n / max(n - 1, 1))
            new_state[f"{pre}.bn{j}.num_batches_tracked"] = sd[f"{pre}.bn{j}.num_batches_tracked"] + 1
        c.update(z1=z1, a1=a1, mask1=mask1, z2=z2, mask2=mask2, bn1=bn1, bn2=bn2, y=y, first_c1=first_c1)
        caches.append(c)
        a_in = y
    return a_in, caches, new_state


def blocks_backward_bf16(dy, P, model_config, caches, rb: Callable = round_bf16):
    """Backward of blocks_forward_bf16 from dy = the (bf16-stored) gradient of the last pooled block output.  Returns the
    gradients of every conv_blocks.* parameter."""
    grads: Dict[str, torch.Tensor] = {}
    nb = len(model_config)
    for i in reversed(range(nb)):
        pre = f"conv_blocks.{i}"
        pool = model_config[i][1]
        c = caches[i]
        w1, w2 = P[f"{pre}.conv1.weight"], P[f"{pre}.conv2.weight"]
        # ---- pool + ReLU + BN2 backward: dz2 = ca*g + cb*z2 + cc (sed_ops.hip: bn_bwd_finalize_kernel) -----------------
        bn2 = c["bn2"]
        g2 = O.avgpool_bwd(dy, pool, c["z2"].shape) * c["mask2"].to(F64)
        xhat2 = (c["z2"] - _c(bn2["mean"])) * _c(bn2["invstd"])
        dz2, dgam2, dbet2 = O.bn_train_bwd(g2, xhat2, P[f"{pre}.bn2.weight"], bn2["invstd"])
        grads[f"{pre}.bn2.weight"], grads[f"{pre}.bn2.bias"] = dgam2, dbet2
        dz2 = rb(dz2)
        grads[f"{pre}.conv2.weight"] = O.conv3x3_wgrad(c["a1"], dz2)
        # ---- conv2 data gradient, ReLU gate, BN1 backward --------------------------------------------------------------
        g1 = rb(O.conv3x3_dgrad(dz2, rb(w2)) * c["mask1"].to(F64))
        bn1 = c["bn1"]
        xhat1 = (c["z1"] - _c(bn1["mean"])) * _c(bn1["invstd"])
        dz1, dgam1, dbet1 = O.bn_train_bwd(g1, xhat1, P[f"{pre}.bn1.weight"], bn1["invstd"])
        grads[f"{pre}.bn1.weight"], grads[f"{pre}.bn1.bias"] = dgam1, dbet1
        if c["first_c1"]:
            # dz1 is never stored: dW1 = ca*A + cb*(w1.G) + cc*sx with A from bf16 operands (g, x), the Gram terms from fp32 x
            n = bn1["n"]
            gam, is_ = P[f"{pre}.bn1.weight"], bn1["invstd"]
            ca = gam * is_
            cb = -gam * is_ * is_ * (dgam1 / n)
            cc = -gam * is_ * (dbet1 / n - bn1["mean"] * is_ * (dgam1 / n))
            A = O.conv3x3_wgrad(rb(c["in1"]), g1)
            Gz = O.conv3x3_wgrad(c["in1"], c["z1"])
            sx = O.conv3x3_wgrad(c["in1"], torch.ones_like(c["z1"]))
            grads[f"{pre}.conv1.weight"] = ca.reshape(-1, 1, 1, 1) * A + cb.reshape(-1, 1, 1, 1) * Gz + cc.reshape(-1, 1, 1, 1) * sx
        else:
            dz1 = rb(dz1)
            grads[f"{pre}.conv1.weight"] = O.conv3x3_wgrad(c["in1"], dz1)
            if i > 0:
                dy = rb(O.conv3x3_dgrad(dz1, rb(w1)))
    return grads


def train_step_grads_bf16(x, target, sd: Dict[str, torch.Tensor], model_config, recall_factor: float,
                          rb: Callable = round_bf16, c1_mode: bool = True):
    """One training step (forward, loss, backward) with the engine's storage rounding.  x: (B, 1, T, F) float32 (already
    z-scored features, as the model receives them).  Returns (loss, logits, grads, new BN running statistics)."""
    P = {k: v.to(F64) for k, v in sd.items()}
    a_in, caches, new_state = blocks_forward_bf16(x, P, sd, model_config, rb, c1_mode)
    ratio = 2 ** O.num_pools_of(model_config)
    logits, hc = O.head_fwd(a_in, P["event_fc.weight"], P["event_fc.bias"], ratio)
    tgt = target.to(F64)
    loss, _ = O.weighted_bce_fwd(logits, tgt, recall_factor)
    dlogits = O.weighted_bce_bwd(logits, tgt, recall_factor)
    da, dW, db = O.head_bwd(dlogits, hc, P["event_fc.weight"], ratio, tuple(a_in.shape))
    grads = blocks_backward_bf16(rb(da), P, model_config, caches, rb)
    grads["event_fc.weight"], grads["event_fc.bias"] = dW, db
    return loss, logits, grads, new_state
