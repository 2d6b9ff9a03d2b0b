"""Counterpart of /root/reference/utils/common.py:11-30 (WeightedBCE); the arithmetic runs in
libsed_hip.so (sed_bce_fwd_bwd)."""
from __future__ import annotations

import torch

from .. import _lib as L
from ..engine import _stream


class _BCEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target, recall_factor):
        B, To, K = output.shape
        dev = output.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dout = torch.empty_like(output)
        scratch = torch.empty(max(1, (B * To * K + 255) // 256), dtype=torch.float32, device=dev)
        L.check(L.lib().sed_bce_fwd_bwd(L.ptr(output), L.ptr(target), L.ptr(loss), L.ptr(dout), L.ptr(scratch), B, To,
                                        K, 1, target.shape[1], float(recall_factor), 1.0, _stream()), "bce_fwd_bwd")
        ctx.save_for_backward(dout)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dout,) = ctx.saved_tensors
        return dout * g, None, None


class WeightedBCE:
    def __init__(self, recall_factor, multi_frame):
        self.recall_factor = float(recall_factor)
        self.multi_frame = multi_frame

    def __call__(self, output, target):
        dev = output.device
        if not output.is_cuda:   # reference eval() hands CPU tensors (train.py:24-26): compute on the GPU anyway
            output = output.cuda()
        target = target.to(output.device)
        if self.multi_frame:
            # (batch, frames, classes); frame counts differ by the pooling floor: the kernel truncates
            # both to N = min(frames) (common.py:20-22)
            o = output.float().contiguous()
            t = target.float().contiguous()
            if o.dim() != 3 or t.dim() != 3 or o.shape[0] != t.shape[0] or o.shape[2] != t.shape[2]:
                raise ValueError(f"expected (B, T, K) output/target, got {tuple(o.shape)} / {tuple(t.shape)}")
        else:
            o = output.float().reshape(1, -1, 1).contiguous()
            t = target.float().reshape(1, -1, 1).contiguous()
            if o.shape != t.shape:
                raise ValueError("output and target sizes differ")
        loss = _BCEFunction.apply(o, t, self.recall_factor)
        return loss if dev == loss.device else loss.to(dev)
