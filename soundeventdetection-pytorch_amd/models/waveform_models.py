"""Drop-in counterpart of /root/reference/models/waveform_models.py (M5 :9-75) whose arithmetic runs in
libsed_hip.so on an MI355X.  Same constructor signature, forward() shapes ((b, 1, frame_size) float32 ->
(b, classes_num) raw logits) and state_dict keys (`conv_block{1..5}.{0,1,3,4}.*`, `fc.{weight,bias}`), and --
because the parameter holders draw from the RNG in nn.Conv1d / nn.Linear's constructor order -- the same seeded
initialisation, so checkpoints interchange with the reference.

There is no CPU path: calling forward() on a CPU module raises.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn as nn

from ..m5_engine import M5_BLOCKS, M5Engine

AUDIO_CHANNELS = 1       # dataset/common_config.py:6 (waveform_configs imports it)
FRAME_SIZE = 31680       # int(48000 * 0.33 * 2), dataset/common_config.py:1-4

DEFAULT_PRECISION = "bf16"


class _Conv1dParams(nn.Module):
    """Parameter holder with nn.Conv1d's names, shapes and default init (same RNG draws)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (kernel_size,), (stride,), (padding,)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_channels * kernel_size)
        nn.init.uniform_(self.bias, -bound, bound)


class _BatchNorm1dParams(nn.Module):
    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _LinearParams(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)


class _M5Function(torch.autograd.Function):
    """Whole-model forward/backward as ONE autograd node over the HIP pipeline."""

    @staticmethod
    def forward(ctx, model, x, *params):
        P = model._tensor_dict()
        B = x.shape[0]
        pad = (-B) % 8
        if pad:
            if model.training:
                raise RuntimeError("M5 on the MI355X interleaves 8 frames per tile: training batches must be a "
                                   "multiple of 8 (BatchNorm statistics must not see padding frames)")
            x = torch.cat([x, x.new_zeros((pad,) + tuple(x.shape[1:]))], 0)
        plan = model.engine.forward(x, P, model.training)
        if model.training:
            model._nbt_pending += 1
        model._fwd_serial += 1
        ctx.model, ctx.plan, ctx.serial, ctx.training = model, plan, model._fwd_serial, model.training
        return plan.pre[:B].clone()

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if not ctx.training:
            raise RuntimeError("backward through an eval-mode forward is not supported; call model.train()")
        if ctx.serial != model._fwd_serial:
            raise RuntimeError("the activations of this forward were overwritten by a later forward of the same shape")
        P = model._tensor_dict()
        names = [n for n, _ in model.named_parameters()]
        G = {n: torch.empty_like(P[n]) for n in names}
        model.engine.backward(ctx.plan, P, G, dlogits=dlogits)
        return (None, None) + tuple(G[n] for n in names)


class M5(nn.Module):
    """Model described in "Very deep convolutional neural networks for raw waveforms" (waveform_models.py:9-12)."""

    def __init__(self, classes_num, precision=None):
        super().__init__()
        self.classes_num = classes_num
        self.precision = precision or DEFAULT_PRECISION
        for name, convs, pooled in M5_BLOCKS:
            mods = []
            for (ci, bi, cin, cout) in convs:
                if cin == 1:
                    mods.append(_Conv1dParams(AUDIO_CHANNELS, cout, kernel_size=79, stride=4, padding=39))
                else:
                    mods.append(_Conv1dParams(cin, cout, kernel_size=3, stride=1, padding=1))
                mods.append(_BatchNorm1dParams(cout))
                mods.append(nn.Identity())          # ReLU slot (fused into the kernels)
            if pooled:
                mods.append(nn.Identity())          # MaxPool1d(4, 4) slot (fused)
            setattr(self, name, nn.Sequential(*mods))
        self.fc = _LinearParams(256, classes_num)
        self.engine = M5Engine(classes_num, self.precision)
        self._fwd_serial = 0
        self._nbt_pending = 0

    def set_precision(self, precision: str):
        self.precision = precision
        self.engine = M5Engine(self.classes_num, precision)
        return self

    def _flush_counters(self):
        if self._nbt_pending:
            n, self._nbt_pending = self._nbt_pending, 0
            for m in self.modules():
                if isinstance(m, _BatchNorm1dParams):
                    m.num_batches_tracked += n

    def state_dict(self, *args, **kwargs):
        self._flush_counters()
        return super().state_dict(*args, **kwargs)

    def _tensor_dict(self) -> Dict[str, torch.Tensor]:
        d = {n: p.data for n, p in self.named_parameters()}
        d.update({n: b for n, b in self.named_buffers()})
        return d

    def forward(self, x):
        # x: (b, c, frame_size) -> (b, classes_num) raw logits
        if not x.is_cuda:
            raise RuntimeError("M5 runs on the MI355X only: move the model and the input to 'cuda' (there is no CPU path)")
        if not next(self.parameters()).is_cuda:
            raise RuntimeError("model parameters are on the CPU; call model.to('cuda')")
        params = [p for _, p in self.named_parameters()]
        return _M5Function.apply(self, x.float(), *params)

    def model_description(self):
        print("Waveform model:")
        n = sum(p.numel() for p in self.parameters() if p.requires_grad)
        print(f"\t- Model has {n} parameters")
