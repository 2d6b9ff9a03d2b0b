"""CPU oracle for the raw-waveform M5 path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file; the
product path (`soundeventdetection-pytorch_amd/`) never does and fails loudly without its HIP library.

Stage-by-stage restatement (explicit forward AND explicit backward formulas, no autograd, no nn.Module)
of what the reference executes on the path

    M5.__init__ / M5.forward            /root/reference/models/waveform_models.py:13-71
    WeightedBCE(multi_frame=False)      /root/reference/utils/common.py:16-30
    Adam(amsgrad)                       /root/reference/train.py:85,101-103

with plain torch CPU tensor ops (float32 = what the reference computes, float64 = a truth to measure both
against).  Layers, in the reference's Sequential indices:

    conv_block1: 0 Conv1d(1, 64, k=79, s=4, p=39, bias)   1 BatchNorm1d   2 ReLU   3 MaxPool1d(4, 4)
    conv_block2: 0 Conv1d(64, 64, k=3, p=1, bias) 1 BN 2 ReLU 3 Conv1d(64, 64) 4 BN 5 ReLU 6 MaxPool1d(4, 4)
    conv_block3: same widths 64 -> 64 -> 64, pooled
    conv_block4: 64 -> 128 -> 128, pooled
    conv_block5: 128 -> 256 -> 256, NOT pooled
    mean over time, fc Linear(256, classes)

Parity status: PINNED.  `tools/gen_golden.py::g7_m5` imports the real `models.waveform_models.M5` in the
build container and writes `tests/golden/g7_m5.npz`; `tests/test_m5_oracle.py` checks this file against it
(logits, loss, every parameter gradient, BN running statistics, parameters after Adam-amsgrad steps).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# (sequential name, [(conv index, bn index, cin, cout, kernel, stride, pad)], pooled)
BLOCKS: List[Tuple[str, List[Tuple[int, int, int, int, int, int, int]], bool]] = [
    ("conv_block1", [(0, 1, 1, 64, 79, 4, 39)], True),
    ("conv_block2", [(0, 1, 64, 64, 3, 1, 1), (3, 4, 64, 64, 3, 1, 1)], True),
    ("conv_block3", [(0, 1, 64, 64, 3, 1, 1), (3, 4, 64, 64, 3, 1, 1)], True),
    ("conv_block4", [(0, 1, 64, 128, 3, 1, 1), (3, 4, 128, 128, 3, 1, 1)], True),
    ("conv_block5", [(0, 1, 128, 256, 3, 1, 1), (3, 4, 256, 256, 3, 1, 1)], False),
]


def layer_list():
    """[(conv prefix, bn prefix, cin, cout, k, stride, pad, pool_after)] in forward order."""
    out = []
    for name, convs, pooled in BLOCKS:
        for i, (ci, bi, cin, cout, k, s, p) in enumerate(convs):
            out.append((f"{name}.{ci}", f"{name}.{bi}", cin, cout, k, s, p, pooled and i == len(convs) - 1))
    return out


def param_names() -> List[str]:
    """model.parameters() order of the reference (waveform_models.py:13-56)."""
    names = []
    for conv, bn, *_ in layer_list():
        names += [conv + ".weight", conv + ".bias", bn + ".weight", bn + ".bias"]
    return names + ["fc.weight", "fc.bias"]


def forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], training: bool, new_state: Dict[str, torch.Tensor] = None):
    """x (B, 1, L) -> logits (B, classes); cache for backward().  waveform_models.py:58-71."""
    cache = []
    a = x
    for conv, bn, cin, cout, k, s, p, pool in layer_list():
        z = F.conv1d(a, sd[conv + ".weight"], sd[conv + ".bias"], stride=s, padding=p)
        g, b = sd[bn + ".weight"], sd[bn + ".bias"]
        if training:
            mean = z.mean(dim=(0, 2))
            var = z.var(dim=(0, 2), unbiased=False)
            if new_state is not None:
                n = z.shape[0] * z.shape[2]
                new_state[bn + ".running_mean"] = (1 - BN_MOMENTUM) * sd[bn + ".running_mean"] + BN_MOMENTUM * mean
                new_state[bn + ".running_var"] = (1 - BN_MOMENTUM) * sd[bn + ".running_var"] + BN_MOMENTUM * var * (n / max(n - 1, 1))
                new_state[bn + ".num_batches_tracked"] = sd[bn + ".num_batches_tracked"] + 1
        else:
            mean, var = sd[bn + ".running_mean"], sd[bn + ".running_var"]
        invstd = torch.rsqrt(var + BN_EPS)
        xhat = (z - mean[None, :, None]) * invstd[None, :, None]
        y = torch.relu(xhat * g[None, :, None] + b[None, :, None])
        idx = None
        if pool:
            yp, idx = F.max_pool1d(y, 4, 4, return_indices=True)
        else:
            yp = y
        cache.append(dict(a=a, xhat=xhat, invstd=invstd, y=y, idx=idx, pool=pool, conv=conv, bn=bn, k=k, s=s, p=p))
        a = yp
    m = a.mean(dim=2)
    logits = m @ sd["fc.weight"].t() + sd["fc.bias"]
    return logits, dict(layers=cache, feat=a, m=m)


def backward(dlogits: torch.Tensor, cache, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Explicit backward of forward(training=True): gradients of every parameter."""
    grads = {}
    grads["fc.weight"] = dlogits.t() @ cache["m"]
    grads["fc.bias"] = dlogits.sum(0)
    dm = dlogits @ sd["fc.weight"]
    L = cache["feat"].shape[2]
    da = (dm / L)[:, :, None].expand(-1, -1, L)
    for c in reversed(cache["layers"]):
        if c["pool"]:
            dy = torch.zeros_like(c["y"])
            dy.scatter_(2, c["idx"], da)          # MaxPool1d backward: the gradient goes to the arg-max
        else:
            dy = da
        dy = dy * (c["y"] > 0)                    # ReLU
        g = sd[c["bn"] + ".weight"]
        n = dy.shape[0] * dy.shape[2]
        grads[c["bn"] + ".bias"] = dy.sum(dim=(0, 2))
        grads[c["bn"] + ".weight"] = (dy * c["xhat"]).sum(dim=(0, 2))
        dz = (g * c["invstd"])[None, :, None] * (dy - grads[c["bn"] + ".bias"][None, :, None] / n
                                                 - c["xhat"] * grads[c["bn"] + ".weight"][None, :, None] / n)
        w = sd[c["conv"] + ".weight"]
        grads[c["conv"] + ".bias"] = dz.sum(dim=(0, 2))           # ~0: BatchNorm removes the bias
        grads[c["conv"] + ".weight"] = torch.nn.grad.conv1d_weight(c["a"], w.shape, dz, stride=c["s"], padding=c["p"])
        if c["a"].shape[1] > 1 or c is not cache["layers"][0]:
            da = torch.nn.grad.conv1d_input(c["a"].shape, w, dz, stride=c["s"], padding=c["p"])
    return grads


def weighted_bce(logits: torch.Tensor, target: torch.Tensor, recall_factor: float):
    """WeightedBCE(multi_frame=False): common.py:26-30.  Returns (loss, dloss/dlogits)."""
    x = logits.reshape(-1)
    y = target.reshape(-1).to(x.dtype)
    w = recall_factor
    loss = -(w * y * F.logsigmoid(x) + (1 - y) * F.logsigmoid(-x)).mean()
    s = torch.sigmoid(x)
    d = (s * (1 + (w - 1) * y) - w * y) / x.numel()
    return loss, d.reshape(logits.shape)


def train_step_grads(x, target, sd, recall_factor: float):
    new_state = {}
    logits, cache = forward(x, sd, True, new_state)
    loss, dlogits = weighted_bce(logits, target, recall_factor)
    return loss, logits, backward(dlogits, cache, sd), new_state


def adam_amsgrad_step(params, grads, state, lr, step, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam(amsgrad=True) single step (train.py:85), in place on `params` / `state`."""
    b1, b2 = betas
    for k, p in params.items():
        g = grads[k]
        m, v, vmax = state.setdefault(k, [torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)])
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        torch.maximum(vmax, v, out=vmax)
        denom = vmax.sqrt() / (1 - b2 ** step) ** 0.5 + eps
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))
