"""Host-side orchestration of the MI355X kernels for the raw-waveform M5 path
(/root/reference/models/waveform_models.py:13-71).

Layout: eight frames are interleaved on the W axis of the conv3x3 kernels' NHWC layout
([N = B/8][L][8][Cp]), so the k=3 Conv1d layers run through the same MFMA kernels as the spectrogram
net (weights expanded to 3x3 with zero side columns: the frames never mix) and BatchNorm1d statistics
over (batch, time) are per-channel reductions over (N, H, W).  Specific to M5: the k=79/stride-4 first
convolution, BN+ReLU+MaxPool1d(4) and its arg-max backward, the mean-over-time + Linear head
(csrc/sed_m5.hip).

Conv1d biases: every convolution feeds a BatchNorm1d, which subtracts the batch mean again, so the
biases cannot influence the output (train or eval) and their true gradient is 0 (the reference's is
fp32 rounding noise).  They are kept in the state_dict, folded into running_mean (which the reference
tracks as mean(z + bias)) and get a zero gradient.

PyTorch is plumbing (device memory, streams, parameter storage); there is no CPU fallback.
"""
from __future__ import annotations

from typing import Dict, List

import os as _os

import torch

from . import _lib as L
from .engine import BN_EPS, BN_MOMENTUM, KernelTimer, _stream

# (block name, [(conv idx, bn idx, cin, cout)], pooled)  -- waveform_models.py:15-56
M5_BLOCKS = [
    ("conv_block1", [(0, 1, 1, 64)], True),
    ("conv_block2", [(0, 1, 64, 64), (3, 4, 64, 64)], True),
    ("conv_block3", [(0, 1, 64, 64), (3, 4, 64, 64)], True),
    ("conv_block4", [(0, 1, 64, 128), (3, 4, 128, 128)], True),
    ("conv_block5", [(0, 1, 128, 256), (3, 4, 256, 256)], False),
]


class _Ly:
    def __init__(self, conv, bn, cin, cout, H, first, pool):
        self.conv, self.bn, self.cin, self.cout, self.H, self.first, self.pool = conv, bn, cin, cout, H, first, pool


class M5Engine:
    def __init__(self, classes_num: int, precision: str = "fp32"):
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.K = int(classes_num)
        self.precision = precision
        self.dt = L.SED_BF16 if precision == "bf16" else L.SED_F32
        self.tdtype = torch.bfloat16 if precision == "bf16" else torch.float32
        self.lib = L.lib()
        self._plans: Dict = {}
        self.timer = None

    def _k(self, name, fn, *args):
        if self.timer is not None:          # engine.KernelTimer: HIP events around the launch (tools/m5_breakdown.py)
            rc = self.timer.launch(name + ":" + getattr(self, "_tag", ""), fn, args)
        else:
            rc = fn(*args)
        L.check(rc, name)

    # ------------------------------------------------------------------------------------------
    def plan(self, B: int, Lw: int, dev):
        key = (B, Lw, str(dev))
        if key in self._plans:
            return self._plans[key]
        if B % 8:
            raise ValueError("the M5 path interleaves 8 frames: the batch must be a multiple of 8 "
                             "(M5.forward pads eval batches itself)")
        lib = self.lib
        N = B // 8
        f32 = dict(dtype=torch.float32, device=dev)
        T = dict(dtype=self.tdtype, device=dev)
        p = type("Plan", (), {})()
        p.B, p.N, p.L = B, N, Lw
        p.layers: List[_Ly] = []
        # "z-free" first block (csrc/sed_m5_mfma.hip, round 4): conv_block1's output is never stored, its three consumers recompute it
        p.zfree = bool(lib.sed_m5_zfree_supported(self.dt))
        # two-pass forward of the first block: statistics without z, then conv + BN + ReLU + MaxPool (+ the z store the backward reads)
        p.fwd2 = p.zfree or bool(lib.sed_m5_fwd2_supported(self.dt))
        # algebraic backward of the first block: statistics + G1 = sum g (x) patch in ONE pass over z, dW1 from G1 and the input's Gram
        # statistics (csrc/sed_m5_mfma.hip); not with the z-free experiment (which has no z to read)
        p.alg = (not p.zfree) and bool(lib.sed_m5_alg_supported(self.dt))
        p.pool_flag = torch.zeros(1, device=dev, dtype=torch.int32)       # sed_maxpool4_pooled_stats raises it, sed_maxpool4_relu_bwd_if resets it
        H = lib.sed_m5_conv1_len(Lw)
        for name, convs, pooled in M5_BLOCKS:
            for i, (ci, bi, cin, cout) in enumerate(convs):
                ly = _Ly(f"{name}.{ci}", f"{name}.{bi}", cin, cout, H, cin == 1, pooled and i == len(convs) - 1)
                ly.z = None if (ly.first and p.zfree) else torch.empty((N, H, 8, cout), **T)
                ly.scale, ly.shift, ly.mean, ly.invstd = (torch.empty(cout, **f32) for _ in range(4))
                ly.coef = torch.empty((3, cout), **f32)
                if ly.first:
                    ly.part = torch.empty((lib.sed_m5_conv1_nparts(B, Lw), 2, cout), **f32)
                else:
                    ly.part = torch.empty((lib.sed_conv_nparts(N, H, 8), 2, cout), **f32)
                    ly.w33 = torch.zeros((cout, cin, 3, 3), **f32)
                    ly.dw33 = torch.empty((cout, cin, 3, 3), **f32)
                    ly.wpack = torch.empty(9 * cin * cout, **T)
                    ly.wpack_t = torch.empty(9 * cin * cout, **T)
                    ly.dwpack = torch.empty(9 * cin * cout, **f32)
                p.layers.append(ly)
            if pooled:
                if H < 4:
                    raise ValueError("frame too short for the MaxPool1d(4) stack")
                Ho = H // 4
            else:
                Ho = H
            last = p.layers[-1]
            last.Ho = Ho
            last.y = torch.empty((N, Ho, 8, last.cout), **T)       # block output (post BN/ReLU/pool)
            last.dy = torch.empty((N, Ho, 8, last.cout), **T)
            H = Ho
        p.t_out = H
        C = p.layers[-1].cout
        p.m = torch.empty((B, C), **f32)
        p.pre = torch.empty((B, self.K), **f32)
        p.dpre = torch.empty((B, self.K), **f32)
        p.loss = torch.zeros(1, **f32)
        p.loss_partial = torch.empty(max(1, (B * self.K + 255) // 256), **f32)
        maxact = max(l.z.numel() for l in p.layers if l.z is not None)
        p.scratch = [torch.empty(maxact, **T) for _ in range(3)]
        p.wgrad_ws = torch.empty(max(1, max(lib.sed_conv_wgrad_ws_floats(N, l.H, 8, l.cin, l.cout) for l in p.layers if not l.first)), **f32)
        p.c1_ws = torch.empty((lib.sed_m5_conv1_nparts(B, Lw), 80, 64), **f32)
        p.c1_dw = torch.empty((80, 64), **f32)
        if p.alg:
            gn = lib.sed_m5_conv1_gram_floats()
            p.gram_part = torch.empty((lib.sed_m5_conv1_nparts(B, Lw), gn), **f32)
            p.gram = torch.empty(gn, **f32)
            p.dw1 = torch.empty((64, 79), **f32)
        nb = max([lib.sed_m5_conv1_nparts(B, Lw)] + [lib.sed_maxpool4_bwd_nparts(N, l.H, 8, l.cout) for l in p.layers if l.pool] +
                 [lib.sed_pool_bwd_nparts(N, l.H, 8, l.cout) for l in p.layers] + [lib.sed_conv_nparts(N, l.H, 8) for l in p.layers])
        p.bwd_part = torch.empty((nb, 2, max(l.cout for l in p.layers)), **f32)
        p.trained = False
        self._plans[key] = p
        return p

    # ------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, P: Dict[str, torch.Tensor], training: bool, update_running_stats: bool = True):
        """x: (B, 1, L) float32 cuda, B % 8 == 0.  Leaves the logits in plan.pre (B, K)."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[1] == 1):
            raise ValueError("expected a float32 CUDA tensor of shape (B, 1, L)")
        x = x.contiguous()
        B, _, Lw = x.shape
        p = self.plan(B, Lw, x.device)
        lib, dt, st, N = self.lib, self.dt, _stream(), p.N
        p.x_ref, p.trained = x, training
        prev = None
        for i, ly in enumerate(p.layers):
            self._tag = f"fwd {ly.conv} {ly.cin}->{ly.cout} L{ly.H}"
            w, bias = P[ly.conv + ".weight"], P[ly.conv + ".bias"]
            g, b = P[ly.bn + ".weight"], P[ly.bn + ".bias"]
            rm, rv = P[ly.bn + ".running_mean"], P[ly.bn + ".running_var"]
            part = ly.part if training else None
            if ly.first and training and p.alg:      # Gram statistics of the input patches (input only: any time before the backward)
                self._k("sed_m5_conv1_gram", lib.sed_m5_conv1_gram, L.ptr(x), L.ptr(p.gram_part), B, Lw, st)
                self._k("sed_sum_partials", lib.sed_sum_partials, L.ptr(p.gram_part), p.gram_part.shape[0], p.gram.numel(), L.ptr(p.gram), st)
            if ly.first and p.fwd2:
                if training:      # BatchNorm statistics of z1 without z1 (eval: the running statistics need no pass at all)
                    self._k("sed_m5_conv1_stats", lib.sed_m5_conv1_stats, dt, L.ptr(x), L.ptr(w), L.ptr(part), B, Lw, st)
            elif ly.first:
                self._k("sed_m5_conv1_fwd", lib.sed_m5_conv1_fwd, dt, L.ptr(x), L.ptr(w), L.ptr(ly.z), L.ptr(part), B, Lw, st)
            else:
                ly.w33[:, :, :, 1].copy_(w)              # Conv1d tap k -> 3x3 tap (k, centre column)
                self._k("sed_pack_conv_weight", lib.sed_pack_conv_weight, dt, L.ptr(ly.w33), L.ptr(ly.wpack), ly.cout, ly.cin,
                        ly.cout, ly.cin, 0, st)
                pl = p.layers[i - 1]
                if pl.pool or pl.first or hasattr(pl, "y"):
                    src, pro, ps, ph = pl.y, L.PRO_NONE, None, None
                else:
                    src, pro, ps, ph = pl.z, L.PRO_BNRELU, pl.scale, pl.shift
                self._k("sed_conv3x3_fwd", lib.sed_conv3x3_fwd_col, dt, pro, L.EPI_STATS if training else L.EPI_STORE, L.ptr(src),
                        L.ptr(ps), L.ptr(ph), L.ptr(ly.wpack), L.ptr(ly.z), None, None, None, None, None, L.ptr(part), N,
                        ly.H, 8, ly.cin, ly.cout, st)
            if training:
                # running_mean of the reference tracks mean(z + bias): take the bias out before the update and
                # put it back after (z here is bias-free); running_var does not depend on it
                if update_running_stats:
                    rm.sub_(bias)
                self._k("sed_bn_train_finalize", lib.sed_bn_train_finalize, L.ptr(ly.part), ly.part.shape[0], float(B * ly.H),
                        L.ptr(g), L.ptr(b), L.ptr(rm) if update_running_stats else None,
                        L.ptr(rv) if update_running_stats else None, BN_MOMENTUM, BN_EPS, L.ptr(ly.scale), L.ptr(ly.shift),
                        L.ptr(ly.mean), L.ptr(ly.invstd), ly.cout, ly.cout, st)
                if update_running_stats:
                    rm.add_(bias)
            else:
                # (z here is bias-free: evaluate against running_mean - bias; a plan-owned buffer, not a temporary whose
                #  storage could be recycled before the kernel runs on another stream / under graph capture)
                if getattr(ly, "rm_nobias", None) is None:
                    ly.rm_nobias = torch.empty_like(rm)
                torch.sub(rm, bias, out=ly.rm_nobias)
                self._k("sed_bn_eval_coeffs", lib.sed_bn_eval_coeffs, L.ptr(g), L.ptr(b), L.ptr(ly.rm_nobias), L.ptr(rv), BN_EPS,
                        L.ptr(ly.scale), L.ptr(ly.shift), ly.cout, ly.cout, st)
            if hasattr(ly, "y"):
                if ly.first and p.fwd2:
                    self._k("sed_m5_conv1_bn_relu_pool_fwd", lib.sed_m5_conv1_bn_relu_pool_fwd, dt, L.ptr(x), L.ptr(w), L.ptr(ly.scale),
                            L.ptr(ly.shift), L.ptr(ly.y), None if p.zfree else (L.ptr(ly.z) if training else None), B, Lw, st)
                elif ly.pool:
                    self._k("sed_bn_relu_maxpool4_fwd", lib.sed_bn_relu_maxpool4_fwd, dt, L.ptr(ly.z), L.ptr(ly.scale),
                            L.ptr(ly.shift), L.ptr(ly.y), N, ly.H, 8, ly.cout, st)
                else:
                    self._k("sed_bn_relu_pool_fwd", lib.sed_bn_relu_pool_fwd, dt, L.ptr(ly.z), L.ptr(ly.scale), L.ptr(ly.shift),
                            L.ptr(ly.y), N, ly.H, 8, ly.cout, 1, st)
        last = p.layers[-1]
        self._k("sed_m5_head_fwd", lib.sed_m5_head_fwd, dt, L.ptr(last.y), L.ptr(P["fc.weight"]), L.ptr(P["fc.bias"]), L.ptr(p.m),
                L.ptr(p.pre), B, p.t_out, last.cout, last.cout, self.K, st)
        return p

    def loss_and_grad(self, p, target: torch.Tensor, recall_factor: float, grad_scale: float = 1.0):
        """WeightedBCE(multi_frame=False) (utils/common.py:26-30); fills plan.dpre; returns plan.loss (1,)."""
        target = target.reshape(p.B, -1).float().contiguous()
        if target.shape[1] != self.K:
            raise ValueError("target must have one label per (frame, class)")
        self._k("sed_bce_fwd_bwd", self.lib.sed_bce_fwd_bwd, L.ptr(p.pre), L.ptr(target), L.ptr(p.loss), L.ptr(p.dpre),
                L.ptr(p.loss_partial), p.B, 1, self.K, 1, 1, float(recall_factor), float(grad_scale), _stream())
        return p.loss

    # ------------------------------------------------------------------------------------------
    def backward(self, p, P: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor], dlogits: torch.Tensor = None,
                 on_group_done=None):
        """Backward of the last training-mode forward; gradients into G[name] (overwritten).  `on_group_done(key)` is
        called when the gradients of a top-level module (fc, conv_block5 ... conv_block1) are enqueued: the
        data-parallel trainer starts that bucket's all-reduce there (train.py: GradAllReducer)."""
        if not p.trained:
            raise RuntimeError("backward() needs a training-mode forward (batch statistics)")
        lib, dt, st, N, B = self.lib, self.dt, _stream(), p.N, p.B
        src = p.dpre if dlogits is None else dlogits.contiguous().float()
        last = p.layers[-1]
        self._k("sed_m5_head_bwd", lib.sed_m5_head_bwd, dt, L.ptr(src), L.ptr(p.m), L.ptr(P["fc.weight"]), L.ptr(G["fc.weight"]),
                L.ptr(G["fc.bias"]), L.ptr(last.dy), B, p.t_out, last.cout, last.cout, self.K, st)
        if on_group_done is not None:
            on_group_done("fc")
        dzA, dzB, gbuf = p.scratch
        for i in reversed(range(len(p.layers))):
            ly = p.layers[i]
            self._tag = f"bwd {ly.conv} {ly.cin}->{ly.cout} L{ly.H}"
            H, C = ly.H, ly.cout
            count = float(B * H)
            gname, bname = ly.bn + ".weight", ly.bn + ".bias"
            G[ly.conv + ".bias"].zero_()          # BatchNorm removes the conv bias: zero gradient
            ca, cb, cc = ly.coef[0], ly.coef[1], ly.coef[2]
            if hasattr(ly, "y"):
                # ---- block output layer: (pool +) ReLU + BN backward statistics from dy -------------------
                if ly.first and p.zfree:
                    # statistics of the pool / ReLU backward with z1 recomputed from the input (one partial row per workgroup)
                    nparts = lib.sed_m5_conv1_nparts(B, p.L)
                    self._k("sed_m5_conv1_pool_bwd_stats", lib.sed_m5_conv1_pool_bwd_stats, dt, L.ptr(p.x_ref), L.ptr(P[ly.conv + ".weight"]),
                            L.ptr(ly.dy), L.ptr(ly.scale), L.ptr(ly.shift), L.ptr(ly.mean), L.ptr(ly.invstd), L.ptr(p.bwd_part), B, p.L, st)
                    dzmode, gsrc, pool = L.DZ_BN, gbuf, 1
                elif ly.first and p.alg:
                    # ONE pass over z1: the pool / ReLU backward statistics and G1 = sum g (x) patch (per-workgroup partials)
                    nparts = lib.sed_m5_conv1_nparts(B, p.L)
                    self._k("sed_m5_conv1_bwd_stats_g1", lib.sed_m5_conv1_bwd_stats_g1, dt, L.ptr(p.x_ref), L.ptr(ly.dy), L.ptr(ly.z),
                            L.ptr(ly.scale), L.ptr(ly.shift), L.ptr(ly.mean), L.ptr(ly.invstd), L.ptr(p.bwd_part), L.ptr(p.c1_ws), B, p.L, st)
                    dzmode, gsrc, pool = L.DZ_BN, gbuf, 1
                elif ly.pool:
                    nparts = lib.sed_maxpool4_bwd_nparts(N, H, 8, C)
                    # first layer, bf16: its matrix-pipe weight gradient rebuilds g from (dy, z) itself -> statistics only here
                    g_free = ly.first and dt == L.SED_BF16 and _os.environ.get("SED_M5_MFMA", "1") != "0"
                    if g_free and _os.environ.get("SED_M5_POOLSTATS", "1") != "0":
                        # statistics from the pooled tensors (y, dy): a quarter of z's rows each; the z pass only runs when an
                        # ill-conditioned channel (|beta| > 8 |gamma|) raised the flag.  ONE flag word: sed_maxpool4_relu_bwd_if resets it
                        # on the stream after consuming it, so the same pointers serve every step (and a graph replay of them)
                        fl = p.pool_flag
                        self._k("sed_maxpool4_pooled_stats", lib.sed_maxpool4_pooled_stats, dt, L.ptr(ly.dy), L.ptr(ly.y), L.ptr(ly.scale),
                                L.ptr(ly.shift), L.ptr(ly.mean), L.ptr(ly.invstd), L.ptr(p.bwd_part), fl.data_ptr(), None, N, H, 8, C, st)
                        self._k("sed_maxpool4_relu_bwd_if", lib.sed_maxpool4_relu_bwd_if, fl.data_ptr(), dt, L.ptr(ly.dy), L.ptr(ly.z),
                                L.ptr(ly.scale), L.ptr(ly.shift), L.ptr(ly.mean), L.ptr(ly.invstd), L.ptr(p.bwd_part), N, H, 8, C, st)
                    else:
                        self._k("sed_maxpool4_relu_bwd", lib.sed_maxpool4_relu_bwd, dt, L.ptr(ly.dy), L.ptr(ly.z), L.ptr(ly.scale),
                                L.ptr(ly.shift), L.ptr(ly.mean), L.ptr(ly.invstd), None if g_free else L.ptr(gbuf), L.ptr(p.bwd_part),
                                N, H, 8, C, st)
                    dzmode, gsrc, pool = L.DZ_BN, gbuf, 1
                else:
                    nparts = lib.sed_pool_bwd_nparts(N, H, 8, C)
                    self._k("sed_pool_relu_bwd_stats", lib.sed_pool_relu_bwd_stats, dt, L.ptr(ly.dy), L.ptr(ly.z), L.ptr(ly.scale),
                            L.ptr(ly.shift), L.ptr(ly.mean), L.ptr(ly.invstd), L.ptr(p.bwd_part), N, H, 8, C, 1, st)
                    dzmode, gsrc, pool = L.DZ_POOL, ly.dy, 1
            else:
                # ---- first conv of a block: g (ReLU-masked data gradient) and its statistics came from the
                #      data-gradient epilogue of the layer above (in dzB / bwd_part) -----------------------------
                nparts = lib.sed_conv_nparts(N, H, 8)
                dzmode, gsrc, pool = L.DZ_BN, dzB, 1
            self._k("sed_bn_bwd_finalize", lib.sed_bn_bwd_finalize, L.ptr(p.bwd_part), nparts, count, L.ptr(P[gname]), L.ptr(ly.mean),
                    L.ptr(ly.invstd), L.ptr(G[gname]), L.ptr(G[bname]), L.ptr(ca), L.ptr(cb), L.ptr(cc), C, C, st)
            if ly.first:
                # dz1 = BN backward of g, then the k=79 weight gradient
                if p.alg:         # dW1 = ca*G1 + cb*(w1 . Gram) + cc*Sp: dz1 is never formed
                    self._k("sed_sum_partials", lib.sed_sum_partials, L.ptr(p.c1_ws), p.c1_ws.shape[0], 80 * 64, L.ptr(p.c1_dw), st)
                    self._k("sed_m5_conv1_wgrad_combine", lib.sed_m5_conv1_wgrad_combine, L.ptr(p.c1_dw), L.ptr(p.gram),
                            L.ptr(P[ly.conv + ".weight"]), L.ptr(ca), L.ptr(cb), L.ptr(cc), L.ptr(p.dw1), st)
                    G[ly.conv + ".weight"].copy_(p.dw1.reshape(64, 1, 79))
                    if on_group_done is not None:
                        on_group_done(ly.conv.split(".")[0])
                    continue
                if p.zfree:       # z1 recomputed from the input inside the weight-gradient kernel as well
                    self._k("sed_m5_conv1_wgrad_fused_pool_x", lib.sed_m5_conv1_wgrad_fused_pool_x, dt, L.ptr(p.x_ref),
                            L.ptr(P[ly.conv + ".weight"]), L.ptr(ly.dy), L.ptr(ly.scale), L.ptr(ly.shift), L.ptr(ca), L.ptr(cb), L.ptr(cc),
                            L.ptr(p.c1_ws), B, p.L, st)
                elif dt == L.SED_BF16 and _os.environ.get("SED_M5_MFMA", "1") != "0":
                    # matrix-pipe kernel, dz rebuilt on load from (g, z): no separate BatchNorm-backward pass, dz never written
                    self._k("sed_m5_conv1_wgrad_fused_pool", lib.sed_m5_conv1_wgrad_fused_pool, dt, L.ptr(p.x_ref), L.ptr(ly.dy),
                            L.ptr(ly.z), L.ptr(ly.scale), L.ptr(ly.shift), L.ptr(ca), L.ptr(cb), L.ptr(cc), L.ptr(p.c1_ws), B, p.L, st)
                else:
                    self._k("sed_bn_bwd_apply", lib.sed_bn_bwd_apply, dt, L.ptr(gsrc), L.ptr(ly.z), L.ptr(ca), L.ptr(cb), L.ptr(cc),
                            L.ptr(dzA), N * H * 8, C, st)
                    self._k("sed_m5_conv1_wgrad", lib.sed_m5_conv1_wgrad, dt, L.ptr(p.x_ref), L.ptr(dzA), L.ptr(p.c1_ws), B, p.L, st)
                self._k("sed_sum_partials", lib.sed_sum_partials, L.ptr(p.c1_ws), p.c1_ws.shape[0], 80 * 64, L.ptr(p.c1_dw), st)
                G[ly.conv + ".weight"].copy_(p.c1_dw[:79].t().reshape(64, 1, 79))
                if on_group_done is not None:
                    on_group_done(ly.conv.split(".")[0])
                continue
            pl = p.layers[i - 1]
            if hasattr(pl, "y"):
                xin, pro, ps, ph = pl.y, L.PRO_NONE, None, None
            else:
                xin, pro, ps, ph = pl.z, L.PRO_BNRELU, pl.scale, pl.shift
            # ---- weight gradient with dz produced on load (and written to dzA for the data gradient) ---------
            self._k("sed_conv3x3_wgrad_fused", lib.sed_conv3x3_wgrad_fused, dt, pro, L.ptr(xin), L.ptr(ps), L.ptr(ph), dzmode,
                    L.ptr(gsrc), L.ptr(ly.z), L.ptr(ly.scale), L.ptr(ly.shift), L.ptr(ca), L.ptr(cb), L.ptr(cc), pool, L.ptr(dzA),
                    L.ptr(ly.dwpack), L.ptr(p.wgrad_ws), N, H, 8, ly.cin, C, st)
            self._k("sed_unpack_conv_wgrad", lib.sed_unpack_conv_wgrad, L.ptr(ly.dwpack), L.ptr(ly.dw33), C, ly.cin, C, ly.cin, st)
            G[ly.conv + ".weight"].copy_(ly.dw33[:, :, :, 1])
            # ---- data gradient --------------------------------------------------------------------------------
            self._k("sed_pack_conv_weight", lib.sed_pack_conv_weight, dt, L.ptr(ly.w33), L.ptr(ly.wpack_t), C, ly.cin, C, ly.cin, 1, st)
            if hasattr(pl, "y"):       # into the pooled block output below: plain store
                self._k("sed_conv3x3_fwd", lib.sed_conv3x3_fwd_col, dt, L.PRO_NONE, L.EPI_STORE, L.ptr(dzA), None, None, L.ptr(ly.wpack_t),
                        L.ptr(pl.dy), None, None, None, None, None, None, N, H, 8, C, ly.cin, st)
            else:                      # into the first conv of this block: fused ReLU mask + BN-backward statistics
                self._k("sed_conv3x3_fwd", lib.sed_conv3x3_fwd_col, dt, L.PRO_NONE, L.EPI_RELUBWD, L.ptr(dzA), None, None,
                        L.ptr(ly.wpack_t), L.ptr(dzB), L.ptr(pl.z), L.ptr(pl.scale), L.ptr(pl.shift), L.ptr(pl.mean),
                        L.ptr(pl.invstd), L.ptr(p.bwd_part), N, H, 8, C, ly.cin, st)
            if on_group_done is not None and hasattr(pl, "y"):      # this was the first conv of its block
                on_group_done(ly.conv.split(".")[0])

    def adam_step(self, flat_p, flat_g, flat_m, flat_v, flat_vmax, lr: float, step: int, grad_scale: float = 1.0,
                  betas=(0.9, 0.999), eps: float = 1e-8):
        self._k("sed_adam_amsgrad_step", self.lib.sed_adam_amsgrad_step, L.ptr(flat_p), L.ptr(flat_g), L.ptr(flat_m),
                L.ptr(flat_v), L.ptr(flat_vmax), flat_p.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
                int(step), float(grad_scale), _stream())
