"""Drop-in counterparts of /root/reference/models/spectogram_models.py (ConvBlock :128-160,
Cnn_AvgPooling :163-230, interpolate :9-22, init_layer/init_bn :25-40) whose arithmetic runs in
libsed_hip.so on an MI355X.  Same constructor signatures, forward() shapes and state_dict keys
(`conv_blocks.{i}.conv{1,2}.weight`, `conv_blocks.{i}.bn{1,2}.{weight,bias,running_mean,
running_var,num_batches_tracked}`, `event_fc.{weight,bias}`), so checkpoints interchange with the
reference's `train.py:123-128` / `main.py:37-39`.

There is no CPU path: calling forward() on a CPU module raises.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn as nn

from .. import _lib as L
from ..engine import CnnEngine, num_pools_of, pad32, _stream

DEFAULT_CHANNEL_AND_POOL = [(64, 2), (128, 2), (256, 2), (512, 1)]   # spectogram_models.py:7
AUDIO_CHANNELS = 1      # dataset/common_config.py:6
MEL_BINS = 64           # dataset/spectogram/spectogram_configs.py:6

# arithmetic precision of newly built models: 'bf16' (bf16 storage + MFMA, fp32 accumulate) or
# 'fp32' (fp32 storage, exact-f32 MFMA: the mode the 1e-3 logit parity gate is judged in)
DEFAULT_PRECISION = "bf16"


def interpolate(x, ratio):
    """(batch, time_steps, classes) -> (batch, time_steps*ratio, classes), each step repeated."""
    if x.is_cuda and x.dtype == torch.float32:
        B, t, K = x.shape
        x = x.contiguous()
        out = torch.empty((B, t * ratio, K), dtype=torch.float32, device=x.device)
        L.check(L.lib().sed_interpolate(L.ptr(x), L.ptr(out), B, t, K, int(ratio), _stream()), "interpolate")
        return out
    return x.repeat_interleave(ratio, dim=1)     # shape utility for host-side tensors


def init_layer(layer, nonlinearity='leaky_relu'):
    nn.init.kaiming_uniform_(layer.weight, nonlinearity=nonlinearity)
    if hasattr(layer, 'bias') and layer.bias is not None:
        layer.bias.data.fill_(0.)


def init_bn(bn):
    bn.bias.data.fill_(0.)
    bn.running_mean.data.fill_(0.)
    bn.weight.data.fill_(1.)
    bn.running_var.data.fill_(1.)


class _Conv3x3Params(nn.Module):
    """Parameter holder with nn.Conv2d(3x3, bias=False)'s weight shape and default init draw."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))   # same RNG draw as nn.Conv2d.__init__
        self.bias = None


class _BatchNormParams(nn.Module):
    """Parameter/buffer holder with nn.BatchNorm2d's names (eps 1e-5, momentum 0.1)."""

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
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))   # same RNG draws as nn.Linear.__init__
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)


class _BlockFunction(torch.autograd.Function):
    """One ConvBlock (conv3x3-BN-ReLU twice, avg-pool) as an autograd node over the HIP kernels, NCHW fp32 at the
    boundary like the reference module (spectogram_models.py:153-160), including the input gradient."""

    @staticmethod
    def forward(ctx, block, x, w1, w2, g1, b1, g2, b2):
        eng = block._engine()
        P = block._tensor_dict()
        training = block.training
        plan = eng.forward(x, P, training)
        if training:
            block.bn1.num_batches_tracked += 1
            block.bn2.num_batches_tracked += 1
        block._fwd_serial += 1
        ctx.block, ctx.plan, ctx.serial, ctx.training = block, plan, block._fwd_serial, training
        yh = plan.y[0]
        B, Ho, Wo, Cp = yh.shape
        y = torch.empty((B, block.conv2.out_channels, Ho, Wo), dtype=torch.float32, device=x.device)
        L.check(L.lib().sed_nhwc_to_nchw(eng.dt, L.ptr(yh), L.ptr(y), B, block.conv2.out_channels, Ho, Wo, Cp, _stream()),
                "nhwc_to_nchw")
        return y

    @staticmethod
    def backward(ctx, dy):
        block, plan = ctx.block, ctx.plan
        if not ctx.training:
            raise RuntimeError("backward through an eval-mode ConvBlock is not supported (BatchNorm batch statistics "
                               "are needed); call block.train()")
        if ctx.serial != block._fwd_serial:
            raise RuntimeError("the activations of this forward were overwritten by a later forward of the same shape")
        eng = block._engine()
        P = block._tensor_dict()
        dy = dy.contiguous().float()
        B, C, Ho, Wo = dy.shape
        L.check(L.lib().sed_nchw_to_nhwc(eng.dt, L.ptr(dy), L.ptr(plan.dy[0]), B, C, Ho, Wo, plan.dy[0].shape[3], _stream()),
                "nchw_to_nhwc")
        names = ["conv1.weight", "conv2.weight", "bn1.weight", "bn1.bias", "bn2.weight", "bn2.bias"]
        G = {"conv_blocks.0." + n: torch.empty_like(P["conv_blocks.0." + n]) for n in names}
        eng.backward(plan, P, G)
        cin = block.conv1.in_channels
        dx = torch.empty((B, cin, plan.T, plan.F), dtype=torch.float32, device=dy.device)
        L.check(L.lib().sed_nhwc_to_nchw(eng.dt, L.ptr(plan.dx), L.ptr(dx), B, cin, plan.T, plan.F, plan.dx.shape[3], _stream()),
                "nhwc_to_nchw")
        return (None, dx) + tuple(G["conv_blocks.0." + n] for n in names)


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, pool_size=2, precision=None):
        super().__init__()
        self.pool_size = pool_size
        self.precision = precision
        self.conv1 = _Conv3x3Params(in_channels, out_channels)
        self.conv2 = _Conv3x3Params(out_channels, out_channels)
        self.bn1 = _BatchNormParams(out_channels)
        self.bn2 = _BatchNormParams(out_channels)
        self.init_weights()
        self._eng = None
        self._fwd_serial = 0

    def init_weights(self):
        init_layer(self.conv1)
        init_layer(self.conv2)
        init_bn(self.bn1)
        init_bn(self.bn2)

    def _engine(self):
        prec = self.precision or DEFAULT_PRECISION
        if self._eng is None or self._eng.precision != prec:
            self._eng = CnnEngine(1, [(self.conv2.out_channels, int(self.pool_size))], self.conv1.in_channels, prec,
                                  head="none", generic_first=True)
        return self._eng

    def _tensor_dict(self) -> Dict[str, torch.Tensor]:
        d = {"conv_blocks.0." + n: p.data for n, p in self.named_parameters()}
        d.update({"conv_blocks.0." + n: b for n, b in self.named_buffers()})
        return d

    def forward(self, input):
        """(B, Cin, H, W) float32 on the GPU -> (B, Cout, H // pool, W // pool): spectogram_models.py:153-160 as ONE
        autograd node (inside Cnn_AvgPooling the blocks run fused, NHWC/bf16 end to end, without this NCHW round trip).
        W must be 8, 16, 32 or 64 (the mel axis of the network's four blocks)."""
        if not input.is_cuda or not self.conv1.weight.is_cuda:
            raise RuntimeError("ConvBlock runs on the MI355X only: move the module and the input to 'cuda' "
                               "(there is no CPU path)")
        x = input.float()
        return _BlockFunction.apply(self, x, self.conv1.weight, self.conv2.weight, self.bn1.weight, self.bn1.bias,
                                    self.bn2.weight, self.bn2.bias)


class _ModelFunction(torch.autograd.Function):
    """Whole-model forward/backward as ONE autograd node over the HIP pipeline."""

    @staticmethod
    def forward(ctx, model, x, *params):
        P = model._tensor_dict()
        training = model.training
        plan = model.engine.forward(x, P, training)
        if training:
            model._nbt_pending += 1
        model._fwd_serial += 1
        ctx.model, ctx.plan, ctx.serial = model, plan, model._fwd_serial
        ctx.training = training
        return model.engine.interpolate(plan)

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if not ctx.training:
            raise RuntimeError("backward through an eval-mode forward is not supported (BatchNorm "
                               "batch statistics are needed); call model.train()")
        if ctx.serial != model._fwd_serial:
            raise RuntimeError("the activations of this forward were overwritten by a later forward of the "
                               "same shape; call backward() before the next forward()")
        P = model._tensor_dict()
        names = [n for n, _ in model.named_parameters()]
        G = {n: torch.empty_like(P[n]) for n in names}
        model.engine.backward(ctx.plan, P, G, dlogits=dlogits.contiguous().float())
        return (None, None) + tuple(G[n] for n in names)


class Cnn_AvgPooling(nn.Module):
    def __init__(self, classes_num, model_config=DEFAULT_CHANNEL_AND_POOL, precision=None):
        super().__init__()
        self.model_config = model_config
        self.classes_num = classes_num
        self.precision = precision or DEFAULT_PRECISION
        self.num_pools = num_pools_of(model_config)
        blocks = [ConvBlock(in_channels=AUDIO_CHANNELS, out_channels=model_config[0][0], pool_size=model_config[0][1])]
        for i in range(1, len(model_config)):
            blocks.append(ConvBlock(in_channels=model_config[i - 1][0], out_channels=model_config[i][0],
                                    pool_size=model_config[i][1]))
        self.conv_blocks = torch.nn.Sequential(*blocks)
        self._build_head(model_config[-1][0], classes_num)
        self.init_weights()
        self.engine = self._make_engine(self.precision)
        self._fwd_serial = 0
        self._nbt_pending = 0     # training forwards not yet added to the num_batches_tracked buffers

    def _flush_counters(self):
        """BatchNorm's num_batches_tracked buffers are bookkeeping only (momentum is fixed): training
        forwards are counted on the host and folded into the 8 buffers when somebody looks at them,
        instead of launching 8 one-element kernels per step."""
        if self._nbt_pending:
            n, self._nbt_pending = self._nbt_pending, 0
            for blk in self.conv_blocks:
                blk.bn1.num_batches_tracked += n
                blk.bn2.num_batches_tracked += n

    def state_dict(self, *args, **kwargs):
        self._flush_counters()
        return super().state_dict(*args, **kwargs)

    def init_weights(self):
        init_layer(self.event_fc)

    def _build_head(self, c_last, classes_num):
        self.event_fc = _LinearParams(c_last, classes_num)

    def _make_engine(self, precision):
        return CnnEngine(self.classes_num, self.model_config, AUDIO_CHANNELS, precision)

    def set_precision(self, precision: str):
        self.precision = precision
        self.engine = self._make_engine(precision)
        return self

    def _tensor_dict(self) -> Dict[str, torch.Tensor]:
        d = {n: p.data for n, p in self.named_parameters()}
        d.update({n: b for n, b in self.named_buffers()})
        return d

    def forward(self, x):
        '''Input: (batch_size, channels_num, times_steps, freq_bins) float32 on the GPU.
        Output: raw logits (batch_size, 8*floor(floor(floor(T/2)/2)/2), classes_num).'''
        if not x.is_cuda:
            raise RuntimeError("Cnn_AvgPooling runs on the MI355X only: move the model and the input to "
                               "'cuda' (there is no CPU path)")
        first = next(self.parameters())
        if not first.is_cuda:
            raise RuntimeError("model parameters are on the CPU; call model.to('cuda')")
        x = x.float()
        params = [p for _, p in self.named_parameters()]
        return _ModelFunction.apply(self, x, *params)

    def logits(self, x):
        return torch.sigmoid(self.forward(x))

    def model_description(self, working_sample_rate=48000, hop_size=15840):
        print("Model description")
        b, w = 'b', MEL_BINS
        h = 60 * working_sample_rate // hop_size
        c = AUDIO_CHANNELS
        print(f"\tInput: ({b}, {c}, {h}, {w})")
        for (c, k) in self.model_config:
            h, w = h // k, w // k
            print(f"\tconv_block -> ({b}, {c}, {h}, {w})")
        print(f"\tmean(dim=3) -> ({b}, {c}, {h})")
        print(f"\ttranspose(1,2) -> ({b}, {h}, {c})")
        print(f"\tFC + sigmoid -> ({b}, {h}, {self.classes_num})")
        num_outputs = h
        h *= 2 ** self.num_pools
        frame_duration = hop_size / working_sample_rate
        print(f"\tinterpolate({2 ** self.num_pools})-> ({b}, {h}, {self.classes_num})")
        print(f"\tModel has {num_outputs} outputs before interpolation, each stands for {2 ** self.num_pools} "
              f"frames or {2 ** self.num_pools * frame_duration:.2f}s")
        n = sum(p.numel() for p in self.parameters() if p.requires_grad)
        print(f"\tModel has {n} parameters")


class _GruParams(nn.Module):
    """Parameter holder with nn.GRU(input, hidden, batch_first=True, bidirectional=True)'s names, shapes
    and default init (U(+-1/sqrt(hidden)), same RNG draw order)."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        k = 1.0 / math.sqrt(hidden_size)
        for sfx in ("", "_reverse"):
            for name, shape in (("weight_ih_l0", (3 * hidden_size, input_size)), ("weight_hh_l0", (3 * hidden_size, hidden_size)),
                                ("bias_ih_l0", (3 * hidden_size,)), ("bias_hh_l0", (3 * hidden_size,))):
                prm = nn.Parameter(torch.empty(shape))
                nn.init.uniform_(prm, -k, k)
                setattr(self, name + sfx, prm)


class Crnn_AvgPooling(Cnn_AvgPooling):
    """BASELINE.json configs[3] (not in the reference repository): the Cnn_AvgPooling
    feature extractor, then mean(dim=3) -> transpose -> bidirectional GRU(C_last, 256) ->
    Linear(512, classes) -> interpolate.  state_dict: conv_blocks.*, gru.{weight_ih_l0,...,
    bias_hh_l0_reverse} (loadable into torch.nn.GRU), event_fc.{weight (classes, 2*hidden), bias}."""

    def __init__(self, classes_num, model_config=DEFAULT_CHANNEL_AND_POOL, precision=None, gru_hidden=256):
        self.gru_hidden = int(gru_hidden)
        super().__init__(classes_num, model_config, precision)

    def _build_head(self, c_last, classes_num):
        self.gru = _GruParams(c_last, self.gru_hidden)
        self.event_fc = _LinearParams(2 * self.gru_hidden, classes_num)

    def _make_engine(self, precision):
        return CnnEngine(self.classes_num, self.model_config, AUDIO_CHANNELS, precision, head="gru",
                         gru_hidden=self.gru_hidden)
