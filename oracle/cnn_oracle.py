"""CPU oracle for the CNN half of the SED training hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file;
the product path (`soundeventdetection-pytorch_amd/`) never does and fails loudly without its HIP
library.

This is a stage-by-stage restatement (explicit forward AND explicit backward formulas, no autograd,
no nn.Module) of what the reference executes on the path

    ConvBlock.forward            /root/reference/models/spectogram_models.py:153-160
    Cnn_AvgPooling.forward       /root/reference/models/spectogram_models.py:185-202
    interpolate                  /root/reference/models/spectogram_models.py:9-22
    init_layer / init_bn         /root/reference/models/spectogram_models.py:25-40
    WeightedBCE.__call__         /root/reference/utils/common.py:16-30
    Adam(amsgrad) + LR decay     /root/reference/train.py:85,101-110

written with plain torch CPU tensor ops so it runs in float32 (what the reference computes) or
float64 (a "truth" to measure both against).  Every intermediate the HIP kernels produce has a
named counterpart in the caches returned here, which is what the parity tests compare.

Parity status: PINNED for these rows.  `tools/gen_golden.py` imports the real reference modules in
the build container and writes `tests/golden/*.npz`; `tests/test_oracle_vs_golden.py` checks this
file against every one of those vectors (forward, autograd gradients, BN running statistics, loss,
Adam-amsgrad trajectories across the LR-decay boundary).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.BatchNorm2d default, spectogram_models.py:142-143
BN_MOMENTUM = 0.1   # nn.BatchNorm2d default


# --------------------------------------------------------------------------------------------
# parameter containers / init (spectogram_models.py:25-40, 163-183)
# --------------------------------------------------------------------------------------------
def kaiming_uniform_leaky_relu_(w: torch.Tensor, generator: Optional[torch.Generator] = None):
    """init_layer (spectogram_models.py:25-31): kaiming_uniform_(nonlinearity='leaky_relu') with the
    default a=0 -> gain sqrt(2); bound = gain*sqrt(3/fan_in); fan_in = Cin*kh*kw (or in_features)."""
    fan_in = w[0].numel()
    bound = math.sqrt(2.0) * math.sqrt(3.0 / fan_in)
    with torch.no_grad():
        w.uniform_(-bound, bound, generator=generator)
    return w


def num_pools_of(model_config: Sequence[Tuple[int, int]]) -> int:
    """Cnn_AvgPooling.__init__ (spectogram_models.py:167-173): starts at 1 REGARDLESS of block 0's
    pool size, +1 for every later block with pool_size == 2."""
    n = 1
    for (_, p) in list(model_config)[1:]:
        if p == 2:
            n += 1
    return n


def make_state(classes_num: int, model_config, in_channels: int = 1, seed: int = 0,
               dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """A state_dict with exactly the reference's key names / shapes, reference init."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    cin = in_channels
    for i, (c, _) in enumerate(model_config):
        for j, (ci, co) in enumerate(((cin, c), (c, c)), start=1):
            sd[f"conv_blocks.{i}.conv{j}.weight"] = kaiming_uniform_leaky_relu_(
                torch.empty(co, ci, 3, 3, dtype=dtype), g)
        for j in (1, 2):
            sd[f"conv_blocks.{i}.bn{j}.weight"] = torch.ones(c, dtype=dtype)
            sd[f"conv_blocks.{i}.bn{j}.bias"] = torch.zeros(c, dtype=dtype)
            sd[f"conv_blocks.{i}.bn{j}.running_mean"] = torch.zeros(c, dtype=dtype)
            sd[f"conv_blocks.{i}.bn{j}.running_var"] = torch.ones(c, dtype=dtype)
            sd[f"conv_blocks.{i}.bn{j}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
        cin = c
    sd["event_fc.weight"] = kaiming_uniform_leaky_relu_(torch.empty(classes_num, cin, dtype=dtype), g)
    sd["event_fc.bias"] = torch.zeros(classes_num, dtype=dtype)
    return sd


PARAM_SUFFIXES = ("conv1.weight", "conv2.weight", "bn1.weight", "bn1.bias", "bn2.weight", "bn2.bias")


def param_names(n_blocks: int) -> List[str]:
    """nn.Module.parameters() order of the reference model: per block conv1.weight, conv2.weight,
    bn1.weight, bn1.bias, bn2.weight, bn2.bias (registration order, spectogram_models.py:132-143),
    then event_fc.weight, event_fc.bias."""
    names = []
    for i in range(n_blocks):
        names += [f"conv_blocks.{i}.{s}" for s in PARAM_SUFFIXES]
    names += ["event_fc.weight", "event_fc.bias"]
    return names


# --------------------------------------------------------------------------------------------
# operators, forward + explicit backward
# --------------------------------------------------------------------------------------------
def conv3x3_fwd(x, w):
    """3x3, stride 1, zero pad 1, no bias, cross-correlation (spectogram_models.py:132-140)."""
    return F.conv2d(x, w, bias=None, stride=1, padding=1)


def conv3x3_dgrad(dy, w):
    """dx[b,c,h,w] = sum_{o,i,j} W[o,c,i,j] dy[b,o,h-i+1,w-j+1] = conv3x3(dy, flip(W)^T)."""
    wt = w.flip(2, 3).transpose(0, 1).contiguous()
    return F.conv2d(dy, wt, bias=None, stride=1, padding=1)


def conv3x3_wgrad(x, dy):
    """dW[o,c,i,j] = sum_{b,h,w} dy[b,o,h,w] x[b,c,h+i-1,w+j-1]."""
    B, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    dw = torch.empty(dy.shape[1], C, 3, 3, dtype=x.dtype)
    for i in range(3):
        for j in range(3):
            dw[:, :, i, j] = torch.einsum("bohw,bchw->oc", dy, xp[:, :, i:i + H, j:j + W])
    return dw


def bn_train_fwd(z, gamma, beta, running_mean, running_var):
    """BatchNorm2d training mode: biased batch variance normalises, unbiased one goes into
    running_var; returns (y, cache, new_running_mean, new_running_var)."""
    n = z.shape[0] * z.shape[2] * z.shape[3]
    mean = z.mean(dim=(0, 2, 3))
    var_b = z.var(dim=(0, 2, 3), unbiased=False)
    invstd = torch.rsqrt(var_b + BN_EPS)
    xhat = (z - mean[None, :, None, None]) * invstd[None, :, None, None]
    y = xhat * gamma[None, :, None, None] + beta[None, :, None, None]
    var_u = var_b * (n / max(n - 1, 1))
    new_rm = (1 - BN_MOMENTUM) * running_mean + BN_MOMENTUM * mean
    new_rv = (1 - BN_MOMENTUM) * running_var + BN_MOMENTUM * var_u
    return y, {"mean": mean, "invstd": invstd, "xhat": xhat}, new_rm, new_rv


def bn_eval_fwd(z, gamma, beta, running_mean, running_var):
    invstd = torch.rsqrt(running_var + BN_EPS)
    return (z - running_mean[None, :, None, None]) * (invstd * gamma)[None, :, None, None] \
        + beta[None, :, None, None]


def bn_train_bwd(dy, xhat, gamma, invstd):
    """dbeta = sum dy; dgamma = sum dy*xhat; dz = gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat))."""
    dbeta = dy.sum(dim=(0, 2, 3))
    dgamma = (dy * xhat).sum(dim=(0, 2, 3))
    n = dy.shape[0] * dy.shape[2] * dy.shape[3]
    dz = (gamma * invstd)[None, :, None, None] * (
        dy - (dbeta / n)[None, :, None, None] - xhat * (dgamma / n)[None, :, None, None])
    return dz, dgamma, dbeta


def avgpool_fwd(a, k):
    """F.avg_pool2d(kernel=stride=k, no pad, floor) (spectogram_models.py:158); k=1 is identity."""
    if k == 1:
        return a
    return F.avg_pool2d(a, kernel_size=k)


def avgpool_bwd(dp, k, in_shape):
    """Spread dy/k^2 over each kxk window; dropped trailing rows/cols get zero."""
    if k == 1:
        return dp
    B, C, H, W = in_shape
    da = torch.zeros(in_shape, dtype=dp.dtype)
    Ho, Wo = H // k, W // k
    up = dp.repeat_interleave(k, dim=2).repeat_interleave(k, dim=3) / float(k * k)
    da[:, :, :Ho * k, :Wo * k] = up
    return da


def conv_block_fwd(x, p: Dict[str, torch.Tensor], prefix: str, pool: int, training: bool,
                   new_state: Optional[Dict[str, torch.Tensor]] = None):
    """ConvBlock.forward (spectogram_models.py:153-160). Returns (pooled, cache)."""
    c: Dict[str, torch.Tensor] = {"x": x}
    a = x
    for j in (1, 2):
        w = p[f"{prefix}.conv{j}.weight"]
        g, b = p[f"{prefix}.bn{j}.weight"], p[f"{prefix}.bn{j}.bias"]
        rm, rv = p[f"{prefix}.bn{j}.running_mean"], p[f"{prefix}.bn{j}.running_var"]
        z = conv3x3_fwd(a, w)
        if training:
            y, bc, nrm, nrv = bn_train_fwd(z, g, b, rm, rv)
            if new_state is not None:
                new_state[f"{prefix}.bn{j}.running_mean"] = nrm
                new_state[f"{prefix}.bn{j}.running_var"] = nrv
                new_state[f"{prefix}.bn{j}.num_batches_tracked"] = \
                    p[f"{prefix}.bn{j}.num_batches_tracked"] + 1
            c[f"mean{j}"], c[f"invstd{j}"], c[f"xhat{j}"] = bc["mean"], bc["invstd"], bc["xhat"]
        else:
            y = bn_eval_fwd(z, g, b, rm, rv)
        c[f"in{j}"] = a
        c[f"z{j}"] = z
        a = torch.relu(y)
        c[f"a{j}"] = a
    out = avgpool_fwd(a, pool)
    c["out"] = out
    return out, c


def conv_block_bwd(dout, c, p, prefix: str, pool: int, need_dx: bool = True):
    """Explicit backward of a training-mode ConvBlock. Returns (dx or None, grads dict)."""
    grads: Dict[str, torch.Tensor] = {}
    da = avgpool_bwd(dout, pool, c["a2"].shape)
    dx = None
    for j in (2, 1):
        g = da * (c[f"a{j}"] > 0).to(da.dtype)                       # relu_ backward
        dz, dgamma, dbeta = bn_train_bwd(g, c[f"xhat{j}"], p[f"{prefix}.bn{j}.weight"], c[f"invstd{j}"])
        grads[f"{prefix}.bn{j}.weight"], grads[f"{prefix}.bn{j}.bias"] = dgamma, dbeta
        grads[f"{prefix}.conv{j}.weight"] = conv3x3_wgrad(c[f"in{j}"], dz)
        c[f"dz{j}"] = dz
        if j == 2 or need_dx:
            da = conv3x3_dgrad(dz, p[f"{prefix}.conv{j}.weight"])
            if j == 1:
                dx = da
    return dx, grads


def interpolate(x, ratio: int):
    """spectogram_models.py:9-22 == repeat_interleave(ratio, dim=1)."""
    return x.repeat_interleave(ratio, dim=1)


def head_fwd(feat, fc_w, fc_b, ratio: int):
    """mean over freq (dim 3) -> transpose -> Linear -> raw logits -> x`ratio` repeat
    (spectogram_models.py:193-200). Returns (interpolated logits (B, t*ratio, K), pre (B,t,K))."""
    m = feat.mean(dim=3).transpose(1, 2)           # (B, t, C)
    pre = m @ fc_w.t() + fc_b
    return interpolate(pre, ratio), {"m": m, "pre": pre}


def head_bwd(dlogits, hc, fc_w, ratio: int, feat_shape):
    B, C, t, Wf = feat_shape
    K = dlogits.shape[2]
    dpre = dlogits.reshape(B, t, ratio, K).sum(dim=2)                 # repeat backward
    dW = torch.einsum("btk,btc->kc", dpre, hc["m"])
    db = dpre.sum(dim=(0, 1))
    dm = dpre @ fc_w                                                   # (B,t,C)
    dfeat = (dm.transpose(1, 2) / Wf)[:, :, :, None].expand(B, C, t, Wf).contiguous()
    return dfeat, dW, db


def weighted_bce_fwd(output, target, recall_factor: float, multi_frame: bool = True):
    """WeightedBCE.__call__ (utils/common.py:16-30): truncate to min frames; mean over all
    elements of -(w*y*log sigma(x) + (1-y)*log sigma(-x))."""
    if multi_frame:
        N = min(output.shape[1], target.shape[1])
        o, t = output[:, :N], target[:, :N]
    else:
        o, t = output.reshape(-1), target
    ls = F.logsigmoid
    loss_el = -(recall_factor * t * ls(o) + (1 - t) * ls(-o))
    return loss_el.mean(), (o, t)


def weighted_bce_bwd(output, target, recall_factor: float):
    """dL/dx = (sigma(x)*(1+(w-1)y) - w*y)/numel over the first N frames; zero beyond."""
    N = min(output.shape[1], target.shape[1])
    o, t = output[:, :N], target[:, :N]
    g = (torch.sigmoid(o) * (1 + (recall_factor - 1) * t) - recall_factor * t) / o.numel()
    full = torch.zeros_like(output)
    full[:, :N] = g
    return full


# --------------------------------------------------------------------------------------------
# whole model
# --------------------------------------------------------------------------------------------
def model_fwd(x, sd, model_config, training: bool, new_state=None):
    """Cnn_AvgPooling.forward. x: (B, 1, T, F). Returns (logits (B,T',K), caches)."""
    caches = []
    a = x
    for i, (_, pool) in enumerate(model_config):
        a, c = conv_block_fwd(a, sd, f"conv_blocks.{i}", pool, training, new_state)
        caches.append(c)
    ratio = 2 ** num_pools_of(model_config)
    logits, hc = head_fwd(a, sd["event_fc.weight"], sd["event_fc.bias"], ratio)
    return logits, {"blocks": caches, "head": hc, "feat_shape": tuple(a.shape), "ratio": ratio}


def model_bwd(dlogits, cache, sd, model_config):
    grads: Dict[str, torch.Tensor] = {}
    da, dW, db = head_bwd(dlogits, cache["head"], sd["event_fc.weight"], cache["ratio"],
                          cache["feat_shape"])
    grads["event_fc.weight"], grads["event_fc.bias"] = dW, db
    for i in reversed(range(len(model_config))):
        da, g = conv_block_bwd(da, cache["blocks"][i], sd, f"conv_blocks.{i}", model_config[i][1],
                               need_dx=(i > 0))
        grads.update(g)
    return grads


def train_step_grads(x, target, sd, model_config, recall_factor: float):
    """forward + loss + backward of one training step; returns (loss, logits, grads, new BN state)."""
    new_state: Dict[str, torch.Tensor] = {}
    logits, cache = model_fwd(x, sd, model_config, training=True, new_state=new_state)
    loss, _ = weighted_bce_fwd(logits, target, recall_factor)
    dlogits = weighted_bce_bwd(logits, target, recall_factor)
    grads = model_bwd(dlogits, cache, sd, model_config)
    return loss, logits, grads, new_state, cache


# --------------------------------------------------------------------------------------------
# Adam-amsgrad (train.py:85) + LR schedule (train.py:108-110)
# --------------------------------------------------------------------------------------------
@dataclass
class AdamState:
    step: int = 0
    m: Dict[str, torch.Tensor] = field(default_factory=dict)
    v: Dict[str, torch.Tensor] = field(default_factory=dict)
    vmax: Dict[str, torch.Tensor] = field(default_factory=dict)


def adam_amsgrad_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor],
                      st: AdamState, lr: float, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam(amsgrad=True, weight_decay=0) single-tensor semantics: the max is taken on
    the UN-bias-corrected v; denom = sqrt(vmax)/sqrt(1-b2^t) + eps; p -= lr/(1-b1^t) * m/denom."""
    st.step += 1
    t = st.step
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    for k, g in grads.items():
        p = params[k]
        if k not in st.m:
            st.m[k] = torch.zeros_like(p)
            st.v[k] = torch.zeros_like(p)
            st.vmax[k] = torch.zeros_like(p)
        st.m[k] = beta1 * st.m[k] + (1 - beta1) * g
        st.v[k] = beta2 * st.v[k] + (1 - beta2) * g * g
        st.vmax[k] = torch.maximum(st.vmax[k], st.v[k])
        denom = st.vmax[k].sqrt() / math.sqrt(bc2) + eps
        params[k] = p - (lr / bc1) * (st.m[k] / denom)


def train_loop(x_batches, y_batches, sd, model_config, recall_factor, lr, num_steps,
               lr_decay_freq: int = 200, lr_decay: float = 0.997):
    """train.py:92-110 without logging: returns (loss trace, final state, final lr)."""
    st = AdamState()
    names = param_names(len(model_config))
    losses = []
    it = 0
    while it < num_steps:
        for xb, yb in zip(x_batches, y_batches):
            loss, _, grads, new_state, _ = train_step_grads(xb, yb, sd, model_config, recall_factor)
            sd.update(new_state)
            adam_amsgrad_step(sd, {k: grads[k] for k in names}, st, lr)
            losses.append(float(loss))
            it += 1
            if it % lr_decay_freq == 0:
                lr *= lr_decay
            if it == num_steps:
                break
    return losses, sd, lr


# --------------------------------------------------------------------------------------------
# decisions used by the "bit-exact" gates (argmax is degenerate for classes_num=1)
# --------------------------------------------------------------------------------------------
def decisions(logits: torch.Tensor) -> torch.Tensor:
    """sigmoid(x) > 0.5  <=>  x > 0."""
    return (logits > 0)


def onset_indices(dec_1d: torch.Tensor) -> torch.Tensor:
    """frame indices where a 0->1 transition happens: flatnonzero(diff(pad(O)) == 1)."""
    d = torch.cat([torch.zeros(1, dtype=torch.int8), dec_1d.to(torch.int8)])
    return torch.nonzero((d[1:] - d[:-1]) == 1).flatten()


# --------------------------------------------------------------------------------------------
# the same step through ATen autograd -- exactly the operator sequence the reference executes
# (nn.Conv2d / nn.BatchNorm2d / relu_ / avg_pool2d / mean / Linear / repeat / BCE-with-logits +
# torch.optim.Adam(amsgrad=True)); used as bench.py's `cpu_baseline` ("port") and cross-checked
# against the explicit formulas above in tests/test_oracle_vs_golden.py.
# --------------------------------------------------------------------------------------------
class AutogradStepper:
    def __init__(self, sd: Dict[str, torch.Tensor], model_config, recall_factor: float, lr: float):
        self.cfg = list(model_config)
        self.names = param_names(len(self.cfg))
        self.params = {k: sd[k].clone().requires_grad_(True) for k in self.names}
        self.buffers = {k: v.clone() for k, v in sd.items() if k not in self.params}
        self.pos_weight = torch.tensor([float(recall_factor)])
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr, betas=(0.9, 0.999), eps=1e-8,
                                    weight_decay=0.0, amsgrad=True)
        self.ratio = 2 ** num_pools_of(self.cfg)
        self.iterations = 0

    def forward(self, x, training=True):
        a = x
        for i, (_, pool) in enumerate(self.cfg):
            for j in (1, 2):
                pre = f"conv_blocks.{i}"
                a = F.conv2d(a, self.params[f"{pre}.conv{j}.weight"], None, 1, 1)
                a = F.batch_norm(a, self.buffers[f"{pre}.bn{j}.running_mean"], self.buffers[f"{pre}.bn{j}.running_var"],
                                 self.params[f"{pre}.bn{j}.weight"], self.params[f"{pre}.bn{j}.bias"], training,
                                 BN_MOMENTUM, BN_EPS)
                a = F.relu_(a)
            a = avgpool_fwd(a, pool)
        m = a.mean(dim=3).transpose(1, 2)
        pre_logits = F.linear(m, self.params["event_fc.weight"], self.params["event_fc.bias"])
        return interpolate(pre_logits, self.ratio)

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
