"""Host-side orchestration of the MI355X kernels for the Cnn_AvgPooling training path.

One `CnnEngine` owns, per input shape, a *plan*: every activation / gradient / workspace buffer the
forward and backward passes need (allocated once through PyTorch-ROCm, laid out NHWC with channels
padded to 32 so that 288 GB of HBM holds whole 60 s batches with all pre-BN conv outputs resident
for the backward pass), and enqueues the libsed_hip.so kernels on the current HIP stream in the
order of

    Cnn_AvgPooling.forward / autograd backward      /root/reference/models/spectogram_models.py:185-202
    WeightedBCE.__call__                            /root/reference/utils/common.py:16-30
    Adam(amsgrad) step                              /root/reference/train.py:85,101-103

PyTorch is plumbing here (device memory, streams, nn.Parameter storage); all arithmetic of the path
happens in the HIP kernels.  There is no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple


import torch

from . import _lib as L

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def num_pools_of(model_config) -> int:
    """spectogram_models.py:167-173 (starts at 1 regardless of block 0)."""
    n = 1
    for (_, p) in list(model_config)[1:]:
        if p == 2:
            n += 1
    return n


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


@dataclass
class _Layer:
    """One conv+BN layer of the plan."""
    cin: int
    cout: int
    cinp: int
    coutp: int
    H: int
    W: int
    z: torch.Tensor = None          # pre-BN conv output (NHWC, dtype)
    part: torch.Tensor = None       # stats partials
    scale: torch.Tensor = None
    shift: torch.Tensor = None
    mean: torch.Tensor = None
    invstd: torch.Tensor = None
    wpack: torch.Tensor = None      # forward operator
    wpack_t: torch.Tensor = None    # data-gradient operator
    dwpack: torch.Tensor = None
    coef: torch.Tensor = None       # [3][Cp] ca, cb, cc


@dataclass
class _Plan:
    B: int
    T: int
    F: int
    layers: List[List[_Layer]] = field(default_factory=list)   # [block][0|1]
    y: List[torch.Tensor] = field(default_factory=list)        # pooled block outputs
    dy: List[torch.Tensor] = field(default_factory=list)
    scratch: List[torch.Tensor] = field(default_factory=list)  # two dz-sized buffers
    m: torch.Tensor = None
    pre: torch.Tensor = None
    dpre: torch.Tensor = None
    loss: torch.Tensor = None
    loss_partial: torch.Tensor = None
    head_ws: torch.Tensor = None
    wgrad_ws: torch.Tensor = None
    c1_ws: torch.Tensor = None
    bwd_part: torch.Tensor = None
    t_out: int = 0
    w_out: int = 0
    x_ref: torch.Tensor = None      # the input of the last forward (first layer wgrad re-reads it)
    trained: bool = False
    gru: dict = None                # buffers of the recurrent head (head == 'gru')


class KernelTimer:
    """HIP-event timing of individual kernel launches on the stream they are enqueued on (torch's
    current stream).  Used by bench.py for the live per-kernel roofline numbers."""

    def __init__(self):
        self.records = []

    def launch(self, label, fn, args):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        self.records.append((label, e0, e1))
        return rc

    def samples(self):
        """label -> [ms of every launch]; call after torch.cuda.synchronize()."""
        out = {}
        for label, e0, e1 in self.records:
            out.setdefault(label, []).append(e0.elapsed_time(e1))
        return out

    def summary(self, outlier_factor: float = 4.0):
        """label -> (launches, total_ms).  An event pair also spans any host stall between the two
        records (the GPU idles while Python is late with the launch), so launches longer than
        `outlier_factor` x the label's median are dropped from both numbers."""
        out = {}
        self.dropped = {}
        for label, ms in self.samples().items():
            srt = sorted(ms)
            med = srt[len(srt) // 2]
            keep = [v for v in ms if v <= outlier_factor * med] if len(ms) >= 3 else ms
            if len(keep) != len(ms):
                self.dropped[label] = len(ms) - len(keep)
            out[label] = (len(keep), sum(keep))
        return out


class CnnEngine:
    def __init__(self, classes_num: int, model_config: Sequence[Tuple[int, int]], in_channels: int = 1,
                 precision: str = "bf16", head: str = "fc", gru_hidden: int = 256, generic_first: bool = False):
        """head: 'fc' (Cnn_AvgPooling), 'gru' (CRNN) or 'none' (a bare stack of ConvBlocks: forward() stops at the last
        pooled output plan.y[-1], backward() takes its gradient).  generic_first: the first conv runs through the general
        Cin >= 1 kernels on an NHWC copy of the (B, Cin, T, F) input and backward() also produces the input gradient
        plan.dx -- the standalone ConvBlock of spectogram_models.py:128-160; the model path keeps the dedicated Cin = 1
        kernels (no input gradient exists there)."""
        if precision not in ("bf16", "fp32", "bf16x3", "f16x3"):
            raise ValueError("precision must be 'bf16', 'fp32', 'f16x3' or 'bf16x3'")
        if head not in ("fc", "gru", "none"):
            raise ValueError("head must be 'fc' (Cnn_AvgPooling), 'gru' (CRNN) or 'none' (ConvBlock stack)")
        if head == "gru" and (gru_hidden % 32 or not 32 <= gru_hidden <= 256):
            raise ValueError("gru_hidden must be a multiple of 32 in [32, 256]")
        self.head, self.Hd = head, int(gru_hidden)
        for (_, p) in model_config:
            if p not in (1, 2):
                raise ValueError("pool sizes must be 1 or 2")
        self.generic_first = bool(generic_first) or in_channels != 1
        self.cin0 = int(in_channels)
        if self.cin0 < 1:
            raise ValueError("in_channels must be >= 1")
        self.K = int(classes_num)
        self.cfg = [(int(c), int(p)) for (c, p) in model_config]
        self.precision = precision
        self.dt = L.SED_BF16 if precision == "bf16" else L.SED_F32
        self.tdtype = torch.bfloat16 if precision == "bf16" else torch.float32
        # "f16x3" / "bf16x3" (round 6): fp32 tensors like "fp32", but the GEMM-shaped launches (operator packing, forward / data gradient,
        # weight gradient) take dtype SED_F32H3 / SED_F32X3 -- every operand split into two 16-bit pieces, three 16-bit MFMAs per product
        # (csrc/sed_conv_x3.hip); every other kernel of the step is the fp32 mode's.  f16x3 (fp16 pieces, ~5e-7 per product) is the fast
        # reference-exact mode; bf16x3 (bf16 pieces, ~1e-5) holds the logit / decision gate but not the fp32 kernels' gradient noise level.
        self.dt_mm = {"bf16x3": L.SED_F32X3, "f16x3": L.SED_F32H3}.get(precision, self.dt)
        self.ratio = 2 ** num_pools_of(self.cfg)
        self._plans: Dict[Tuple[int, int, int, str], _Plan] = {}
        self.lib = L.lib()
        self.timer: Optional[KernelTimer] = None   # set to a KernelTimer to time every launch
        self._tag = ""
        # SyncBN (optional, data parallel): an object with .world and .all_reduce(tensor) (in-place SUM over the ranks, ordered
        # on the current stream).  Every BatchNorm then normalises with the statistics of the GLOBAL batch, like the
        # reference's single process does (spectogram_models.py:142-143 under train.py:95-97): the per-workgroup partial
        # sums are reduced to one row per rank, summed over the ranks, and finalized with count * world.
        self.bn_sync = None
        self.wg_flush_per_group = False      # set by FusedTrainer under data parallel (see backward: group_done)

    def _grad_dtype(self, B: int, H: int, W: int) -> int:
        """dtype argument of a GEMM-shaped BACKWARD launch.  f16x3: fp16 pieces have five exponent bits and a loss gradient is ~1/(B*H*W)
        per element (mean-reduced BCE spread over the layer's pixels), so the streamed gradient operand is scaled by 2^e before the
        split, e = round(log2(B*H*W)) + 2 (the kernel scales the result back; bits 8..15 of dtype, include/sed_hip.h).  fp16's normal
        range leaves ~13 binades either side of that estimate."""
        if self.dt_mm != L.SED_F32H3:
            return self.dt_mm
        import math
        e = max(-100, min(100, int(round(math.log2(max(1, B * H * W)))) + 2))
        return L.SED_F32H3 | ((e & 0xff) << 8)

    def _sync_row(self, part, nparts: int, n: int, out):
        """this rank's partial rows [nparts][n] -> out[n] = sum over rows and over ranks"""
        self._k("sed_sum_partials", self.lib.sed_sum_partials, L.ptr(part), nparts, n, L.ptr(out), _stream())
        self.bn_sync.all_reduce(out)
        return out

    def _k(self, name, fn, *args):
        if self.timer is not None:
            rc = self.timer.launch(f"{name}:{self._tag}" if self._tag else name, fn, args)
        else:
            rc = fn(*args)
        L.check(rc, name)

    # ------------------------------------------------------------------------------------------
    def plan(self, B: int, T: int, F: int, device) -> _Plan:
        key = (B, T, F, str(device))
        if key in self._plans:
            return self._plans[key]
        if F not in (8, 16, 32, 64) and len(self.cfg) > 0:
            pass
        lib = self.lib
        dev = device
        p = _Plan(B, T, F)
        H, W, cin = T, F, self.cin0
        f32 = dict(dtype=torch.float32, device=dev)
        maxact = 0
        max_wgrad_ws = 0
        max_bwd_parts = 0
        for bi, (c, pool) in enumerate(self.cfg):
            if W not in (8, 16, 32, 64):
                raise ValueError(f"mel width {W} at block {bi} unsupported (need 8/16/32/64)")
            if pool == 2 and (H < 2 or W < 2):
                raise ValueError("input too short for the pooling stack")
            blk = []
            for j, (ci, co) in enumerate(((cin, c), (c, c))):
                first = (bi == 0 and j == 0 and not self.generic_first)     # the dedicated Cin = 1 kernels
                cinp = 1 if first else pad32(ci)
                ly = _Layer(ci, co, cinp, pad32(co), H, W)
                nparts = lib.sed_conv_c1_nparts(B, H, W) if first else lib.sed_conv_nparts(B, H, W)
                ly.z = torch.empty((B, H, W, ly.coutp), dtype=self.tdtype, device=dev)
                ly.part = torch.empty((nparts, 2, ly.coutp), **f32)
                ly.scale = torch.empty(ly.coutp, **f32)
                ly.shift = torch.empty(ly.coutp, **f32)
                ly.mean = torch.empty(ly.coutp, **f32)
                ly.invstd = torch.empty(ly.coutp, **f32)
                ly.coef = torch.empty((3, ly.coutp), **f32)
                if not first:
                    ly.wpack = torch.empty(9 * ly.cinp * ly.coutp, dtype=self.tdtype, device=dev)
                    ly.dwpack = torch.empty(9 * ly.cinp * ly.coutp, **f32)
                    max_wgrad_ws = max(max_wgrad_ws, lib.sed_conv_wgrad_ws_floats(B, H, W, ly.cinp, ly.coutp))
                    ly.wpack_t = torch.empty(9 * ly.cinp * ly.coutp, dtype=self.tdtype, device=dev)
                else:
                    ly.dwpack = torch.empty(9 * ly.coutp, **f32)
                maxact = max(maxact, B * H * W * ly.coutp)
                max_bwd_parts = max(max_bwd_parts, lib.sed_pool_bwd_nparts(B, H, W, ly.coutp) * 2 * ly.coutp,
                                    nparts * 2 * ly.coutp, lib.sed_conv_nparts(B, H, W) * 2 * pad32(ci))
                blk.append(ly)
            p.layers.append(blk)
            Ho, Wo = H // pool, W // pool
            p.y.append(torch.empty((B, Ho, Wo, pad32(c)), dtype=self.tdtype, device=dev))
            p.dy.append(torch.empty((B, Ho, Wo, pad32(c)), dtype=self.tdtype, device=dev))
            H, W, cin = Ho, Wo, c
        if H < 1:
            raise ValueError("input too short for the pooling stack")
        p.t_out, p.w_out = H, W
        Cl, Clp = cin, pad32(cin)
        p.m = torch.empty((B, H, Clp), **f32)
        p.pre = torch.empty((B, H, self.K), **f32)
        p.dpre = torch.empty((B, H, self.K), **f32)
        p.loss = torch.zeros(1, **f32)
        p.loss_partial = torch.empty(max(1, (B * H * self.ratio * self.K + 255) // 256), **f32)
        Cfc = 2 * self.Hd if self.head == "gru" else Cl
        p.head_ws = torch.empty(max(1, lib.sed_head_bwd_ws_floats(B, H, Cfc, max(1, self.K))), **f32)
        if self.head == "gru":
            p.gru = self._plan_gru(B, H, Cl, dev)
        p.wgrad_ws = torch.empty(max(1, max_wgrad_ws), **f32)
        l0 = p.layers[0][0]
        p.c1_ws = torch.empty((lib.sed_conv_c1_nparts(B, T, F), 9, l0.coutp), **f32)
        p.c1_gram = torch.empty((lib.sed_conv_c1_gram_nparts(B, T, F), 54), **f32)
        # "C1 mode": block 0 without conv1's output in memory (csrc/conv_common.h): conv2's forward and weight gradient
        # rebuild relu(bn1(conv1(x))) on the matrix pipe from the 1-channel input, the data gradient gates with a bit
        # mask of conv1's ReLU decisions, BN1's statistics and backward come from Gram statistics of the input patches.
        # Removes 4 of block 0's 12.75 HBM passes (6.41 -> 6.2 ms/step); SED_C1_MODE=0 restores the z1 dataflow.
        import os as _os
        p.c1_mode = bool(lib.sed_c1_mode_supported(self.dt, F, self.cfg[0][0], self.cfg[0][0])) and \
            _os.environ.get("SED_C1_MODE", "1") != "0" and self.cfg[0][1] in (1, 2) and not self.generic_first
        if self.generic_first:
            c0p = pad32(self.cin0)
            p.x_nhwc = torch.zeros((B, T, F, c0p), dtype=self.tdtype, device=dev)
            p.dx = torch.empty((B, T, F, c0p), dtype=self.tdtype, device=dev)
        p.c1_A = torch.empty((9, l0.coutp), **f32)
        # fused data gradient of block 0 (csrc/sed_dgrad_c1.hip): per-workgroup partials and sums of [A (9 taps); sum g]
        p.c1_dg_fused = p.c1_mode and _os.environ.get("SED_DGRAD_FUSED", "1") != "0"
        p.c1_a10_part = torch.empty((lib.sed_conv_dgrad_c1_nparts(), 10, 32), **f32)
        p.c1_a10 = torch.empty((10, 32), **f32)
        p.c1_mask = None          # C1 mode: conv1's ReLU decisions as a bit mask -- only when the backward does not derive them (below)
        # Pool + ReLU + BN2 backward statistics from POOLED tensors (include/sed_hip.h, sed_conv3x3_dgrad_poolstats): a 2x2-pooled
        # block whose output gradient comes from the next block's conv1 data gradient gets its statistics in that kernel's
        # epilogue (forward: active-pixel counts beside the pooled activation) -- no separate pass over z2.  SED_POOL_STATS=z
        # keeps the per-pixel pass everywhere.
        nb = len(self.cfg)
        p.pool_fused, p.pool_cnt, p.pool_nparts = [False] * nb, [None] * nb, [0] * nb
        if _os.environ.get("SED_POOL_STATS", "p") != "z":
            for bi in range(nb - 1):
                l2, n1 = p.layers[bi][1], p.layers[bi + 1][0]
                if self.cfg[bi][1] == 2 and lib.sed_dgrad_poolstats_supported(self.dt, n1.W, n1.coutp, n1.cinp) and n1.cinp == l2.coutp:
                    p.pool_fused[bi] = True
                    p.pool_cnt[bi] = torch.empty((B, n1.H, n1.W, l2.coutp), dtype=torch.uint8, device=dev)
                    p.pool_nparts[bi] = lib.sed_conv_nparts(B, n1.H, n1.W)       # (the conditional per-pixel pass adapts its grid)
                    max_bwd_parts = max(max_bwd_parts, p.pool_nparts[bi] * 2 * l2.coutp)
        # Fused weight + data gradient (csrc/sed_bwd_fused.hip): dz of a layer lives only in LDS.  SED_BWD_FUSED=0 restores the
        # two-kernel backward (weight gradient writes dz, the data-gradient launch reads it back).
        p.bwd_fused = [[False, False] for _ in range(nb)]
        if self.precision == "bf16" and _os.environ.get("SED_BWD_FUSED", "1") != "0":
            for bi in range(nb):
                l1, l2 = p.layers[bi]
                if not (bi == 0 and not self.generic_first):
                    epi1 = L.EPI_POOLSTATS if (bi > 0 and p.pool_fused[bi - 1]) else L.EPI_STORE
                    if bi > 0 or self.generic_first:
                        p.bwd_fused[bi][0] = bool(lib.sed_conv3x3_bwd_fused_supported(self.dt, l1.W, l1.cinp, l1.coutp, L.DZ_BN, L.PRO_NONE, epi1))
                # conv2's fused form produces dz from the block's (pooled) output gradient: the library answers per pooling size
                # (W = 32: one dy item per 2x2 window, pool 2 only -- pool-1 blocks keep the two-kernel backward)
                if not (p.c1_mode and bi == 0):
                    p.bwd_fused[bi][1] = bool(lib.sed_conv3x3_bwd_fused_supported_pool(self.dt, l2.W, l2.cinp, l2.coutp, L.DZ_POOL, L.PRO_BNRELU,
                                                                                        L.EPI_RELUBWD, self.cfg[bi][1]))
        # block 0 in C1 mode: weight gradient + fused data gradient of conv2 in one launch (csrc/sed_bwd_fused_c1.hip)
        p.c1_bwd_fused = bool(p.c1_mode and p.c1_dg_fused and self.precision == "bf16" and _os.environ.get("SED_BWD_FUSED", "1") != "0"
                              and lib.sed_conv3x3_bwd_fused_c1_supported(self.dt, F, self.cfg[0][0], self.cfg[0][1]))
        # Round 5: the fused block-0 backward derives conv1's ReLU gate from the activation tile it rebuilds for its weight gradient, so
        # the forward neither builds nor stores the bit mask (SED_C1_GATE=mask: the round-4 form, for the A/B); the unfused kernels
        # (sed_conv3x3_dgrad_c1_stats / sed_conv3x3_dgrad_c1) still take the forward's mask
        # Round 5: block 0's conv1 backward tail (partial sums of [A; sum g] -> BN1 backward coefficients -> dW1 combine) in one launch with
        # the forward's reduced Gram statistics (sed_c1_bwd_tail); SED_C1_TAIL=0 keeps the three kernels (and SyncBN always does)
        p.c1_gsum = torch.empty(54, dtype=torch.float64, device=dev)
        p.c1_tail = bool(p.c1_mode and p.c1_dg_fused and not self.generic_first and _os.environ.get("SED_C1_TAIL", "1") != "0")
        p.c1_gate_derived = bool(p.c1_bwd_fused and _os.environ.get("SED_C1_GATE", "derived") != "mask")
        if p.c1_mode and not p.c1_gate_derived:
            p.c1_mask = torch.empty((B, T, F, 2), dtype=torch.int16, device=dev)
        # Round 6, SED_WGRAD_REDUCE=batch: the weight-gradient reductions of a step in ONE launch at the end of the backward (every layer then
        # keeps its own slab workspace, ~0.25 GB at the bench shape).  Bit-identical gradients; measured NEUTRAL in an interleaved A/B
        # (4.0922 vs 4.0933 ms/step, profiles/r06_l_ab_wgrad_reduce_batch.txt: the seven 10 us reductions are bandwidth, and the launch
        # boundaries around them already overlap the neighbouring kernels' ramps) -- so the default stays one reduction per layer.
        p.wg_defer = _os.environ.get("SED_WGRAD_REDUCE", "inline") == "batch"
        p.wg_pending, p.wg_tables = [], {}
        if p.wg_defer:
            for blk in p.layers:
                for ly in blk:
                    if ly.cinp != 1:
                        ly.wgrad_ws = torch.empty(max(1, lib.sed_conv_wgrad_ws_floats(B, ly.H, ly.W, ly.cinp, ly.coutp)), **f32)
        p.pool_flag = torch.zeros(nb, dtype=torch.int32, device=dev)
        p.bwd_part = torch.empty(max(1, max_bwd_parts), **f32)
        maxc = max(ly.coutp for blk in p.layers for ly in blk)
        p.sync_row = torch.empty(2 * maxc, **f32)          # SyncBN: one all-reduced row of (sum, sum-of-squares) / backward sums
        p.sync_gram = torch.empty(54, **f32)
        p.sync_a10 = torch.empty((10, 32), **f32)
        p.scratch = [torch.empty(maxact, dtype=self.tdtype, device=dev) for _ in range(2)]
        self._plans[key] = p
        return p


    def _plan_gru(self, B, t, C, dev):
        """Buffers of the recurrent head; R = B*t rows, Rp = R padded to 4 (16-byte aligned GEMM rows)."""
        lib, Hd = self.lib, self.Hd
        if C % 4:
            raise ValueError("the GRU head needs a channel count that is a multiple of 4")
        f32 = dict(dtype=torch.float32, device=dev)
        R = B * t
        Rp = (R + 3) // 4 * 4
        g = dict(R=R, Rp=Rp)
        g["m"] = torch.empty((R, C), **f32)
        g["gi"] = torch.empty((R, 6 * Hd), **f32)
        g["hseq"] = torch.empty((R, 2 * Hd), **f32)
        g["saved"] = torch.empty((R, 8 * Hd), **f32)
        g["fc_m"] = torch.empty((R, 2 * Hd), **f32)
        g["dhseq"] = torch.empty((R, 2 * Hd), **f32)
        g["dgi"] = torch.empty((R, 6 * Hd), **f32)
        g["dgh"] = torch.empty((R, 6 * Hd), **f32)
        g["dgiT"] = torch.zeros((6 * Hd, Rp), **f32)
        g["dghT"] = torch.zeros((6 * Hd, Rp), **f32)
        g["mT"] = torch.zeros((C, Rp), **f32)
        g["hprevT"] = torch.zeros((2 * Hd, Rp), **f32)
        g["wihT"] = torch.empty((C, 6 * Hd), **f32)
        g["dm"] = torch.empty((R, C), **f32)
        g["bhh"] = torch.empty((2, 3 * Hd), **f32)
        n = lib.sed_gru_pack_elems(Hd)
        g["pack_f"] = torch.empty(n, dtype=self.tdtype, device=dev)
        g["pack_b"] = torch.empty(n, dtype=self.tdtype, device=dev)
        # split-K of the weight-gradient GEMMs (K = B*t rows): SED_GRU_KSPLIT, read here -- when the plan is built, like the other engine-side
        # SED_* knobs -- and kept with the plan, because the workspace below is sized for it
        import os as _os
        g["ksplit"] = max(1, int(_os.environ.get("SED_GRU_KSPLIT", "64")))
        ws = max(lib.sed_gemm_nt_ws_floats(3 * Hd, C, g["ksplit"]), lib.sed_gemm_nt_ws_floats(3 * Hd, Hd, g["ksplit"]),
                 lib.sed_gemm_tn_ws_floats(3 * Hd, C, g["ksplit"]), lib.sed_gemm_tn_ws_floats(3 * Hd, Hd, g["ksplit"]))
        g["ws"] = torch.empty(max(1, ws), **f32)
        # the four weight-gradient products in one launch (sed_gemm_tn_batch) need a workspace each.  SED_GRU_TN_BATCH=0: four calls (A/B).
        g["tn_batch"] = _os.environ.get("SED_GRU_TN_BATCH", "1") != "0"
        g["ws4"] = [torch.empty(max(1, ws), **f32) for _ in range(4)] if g["tn_batch"] else None
        g["tn_desc"] = (L.GemmTnDesc * 4)()
        # Round 6: the BPTT tail's weight / bias gradients straight from the row-major gate gradients (sed_gemm_tn: reduction index on the
        # rows, shifted hidden-state rows, column sums as a by-product) -- no transposes, no separate bias sums.  SED_GRU_TN=0: the
        # transpose + sed_gemm_nt + sed_row_sums form (A/B).
        g["tn"] = _os.environ.get("SED_GRU_TN", "1") != "0"
        return g

    GRU_DIRS = ("", "_reverse")

    def _gru_forward(self, p, P, feat, training):
        lib, dt, st, Hd, g = self.lib, self.dt, _stream(), self.Hd, p.gru
        B, t = p.B, p.t_out
        Cl = self.cfg[-1][0]
        R = g["R"]
        self._tag = "gru"
        self._k("sed_mel_mean_fwd", lib.sed_mel_mean_fwd, dt, L.ptr(feat), L.ptr(g["m"]), R, p.w_out, Cl, pad32(Cl), st)
        for d, sfx in enumerate(self.GRU_DIRS):
            self._k("sed_gemm_nt", lib.sed_gemm_nt, dt, L.ptr(g["m"]), Cl, L.ptr(P["gru.weight_ih_l0" + sfx]), Cl,
                    L.ptr(P["gru.bias_ih_l0" + sfx]), g["gi"].data_ptr() + 4 * d * 3 * Hd, 6 * Hd, R, 3 * Hd, Cl, 1, None, st)
            g["bhh"][d].copy_(P["gru.bias_hh_l0" + sfx])
        self._k("sed_gru_pack_weights", lib.sed_gru_pack_weights, dt, L.ptr(P["gru.weight_hh_l0"]),
                L.ptr(P["gru.weight_hh_l0_reverse"]), L.ptr(g["pack_f"]), L.ptr(g["pack_b"]), Hd, st)
        self._k("sed_gru_seq_fwd", lib.sed_gru_seq_fwd, dt, L.ptr(g["gi"]), L.ptr(g["bhh"]), L.ptr(g["pack_f"]),
                L.ptr(g["hseq"]), L.ptr(g["saved"]) if training else None, B, t, Hd, st)
        self._k("sed_head_fwd", lib.sed_head_fwd, L.SED_F32, L.ptr(g["hseq"]), L.ptr(P["event_fc.weight"]),
                L.ptr(P["event_fc.bias"]), L.ptr(g["fc_m"]), L.ptr(p.pre), B, t, 1, 2 * Hd, 2 * Hd, self.K, st)
        self._tag = ""

    def _gru_backward(self, p, P, G, src, ratio, dfeat, on_group_done):
        lib, dt, st, Hd, g = self.lib, self.dt, _stream(), self.Hd, p.gru
        B, t = p.B, p.t_out
        Cl = self.cfg[-1][0]
        R, Rp = g["R"], g["Rp"]
        self._tag = "gru"
        self._k("sed_head_bwd", lib.sed_head_bwd, L.SED_F32, L.ptr(src), L.ptr(g["fc_m"]), L.ptr(P["event_fc.weight"]),
                L.ptr(G["event_fc.weight"]), L.ptr(G["event_fc.bias"]), L.ptr(g["dhseq"]), L.ptr(p.head_ws), B, t, 1,
                2 * Hd, 2 * Hd, self.K, ratio, st)
        if on_group_done is not None:
            on_group_done("event_fc")
        self._k("sed_gru_seq_bwd", lib.sed_gru_seq_bwd, dt, L.ptr(g["dhseq"]), L.ptr(g["hseq"]), L.ptr(g["saved"]),
                L.ptr(g["pack_b"]), L.ptr(g["dgi"]), L.ptr(g["dgh"]), B, t, Hd, st)
        ks = max(1, min(g["ksplit"], R // 256))
        if g["tn"] and g["tn_batch"]:
            # dW_ih = dgi_d^T . m (+ db_ih = column sums of dgi_d);  dW_hh = dgh_d^T . h_prev (+ db_hh), both directions: ONE product
            # launch and ONE reduction launch (twelve launches as four sed_gemm_tn calls)
            ds = g["tn_desc"]
            for d, sfx in enumerate(self.GRU_DIRS):
                for j, (a, b, ldb, wname, bname, n, shift) in enumerate((
                        (g["dgi"].data_ptr() + 4 * d * 3 * Hd, L.ptr(g["m"]), Cl, "gru.weight_ih_l0", "gru.bias_ih_l0", Cl, 0),
                        (g["dgh"].data_ptr() + 4 * d * 3 * Hd, g["hseq"].data_ptr() + 4 * d * Hd, 2 * Hd, "gru.weight_hh_l0", "gru.bias_hh_l0", Hd,
                         1 if d == 0 else -1))):
                    e = ds[2 * d + j]
                    e.A, e.B, e.C, e.colsum = a, b, L.ptr(G[wname + sfx]), L.ptr(G[bname + sfx])
                    e.workspace = L.ptr(g["ws4"][2 * d + j]) if ks > 1 else None
                    e.lda, e.ldb, e.ldc, e.M, e.N, e.K, e.seq, e.shift, e.ksplit = 6 * Hd, ldb, n, 3 * Hd, n, R, t, shift, ks
            import ctypes as _C
            self._k("sed_gemm_tn_batch", lib.sed_gemm_tn_batch, dt, _C.cast(ds, _C.c_void_p), 4, st)
            for d, sfx in enumerate(self.GRU_DIRS):
                self._k("sed_transpose_shift", lib.sed_transpose_shift, L.ptr(P["gru.weight_ih_l0" + sfx]), Cl,
                        g["wihT"].data_ptr() + 4 * d * 3 * Hd, 6 * Hd, 3 * Hd, Cl, 3 * Hd, 0, st)
        elif g["tn"]:
            for d, sfx in enumerate(self.GRU_DIRS):
                ws = L.ptr(g["ws"]) if ks > 1 else None
                # dW_ih = dgi_d^T . m (+ db_ih = column sums of dgi_d);  dW_hh = dgh_d^T . h_prev (+ db_hh): h_prev = hseq shifted by one
                # step inside each clip's sequence (forward direction looks one step back, reverse one step ahead)
                self._k("sed_gemm_tn", lib.sed_gemm_tn, dt, g["dgi"].data_ptr() + 4 * d * 3 * Hd, 6 * Hd, L.ptr(g["m"]), Cl,
                        L.ptr(G["gru.weight_ih_l0" + sfx]), Cl, L.ptr(G["gru.bias_ih_l0" + sfx]), 3 * Hd, Cl, R, t, 0, ks, ws, st)
                self._k("sed_gemm_tn", lib.sed_gemm_tn, dt, g["dgh"].data_ptr() + 4 * d * 3 * Hd, 6 * Hd, g["hseq"].data_ptr() + 4 * d * Hd,
                        2 * Hd, L.ptr(G["gru.weight_hh_l0" + sfx]), Hd, L.ptr(G["gru.bias_hh_l0" + sfx]), 3 * Hd, Hd, R, t,
                        1 if d == 0 else -1, ks, ws, st)
                self._k("sed_transpose_shift", lib.sed_transpose_shift, L.ptr(P["gru.weight_ih_l0" + sfx]), Cl,
                        g["wihT"].data_ptr() + 4 * d * 3 * Hd, 6 * Hd, 3 * Hd, Cl, 3 * Hd, 0, st)
        else:
            # (round-4 form) transposes so that the B*t reduction axis is contiguous
            self._k("sed_transpose_shift", lib.sed_transpose_shift, L.ptr(g["dgi"]), 6 * Hd, L.ptr(g["dgiT"]), Rp, R, 6 * Hd, R, 0, st)
            self._k("sed_transpose_shift", lib.sed_transpose_shift, L.ptr(g["dgh"]), 6 * Hd, L.ptr(g["dghT"]), Rp, R, 6 * Hd, R, 0, st)
            self._k("sed_transpose_shift", lib.sed_transpose_shift, L.ptr(g["m"]), Cl, L.ptr(g["mT"]), Rp, R, Cl, R, 0, st)
        for d, sfx in enumerate(self.GRU_DIRS if not g["tn"] else ()):
            # h_prev of every step, transposed: forward direction looks one step back, reverse one step ahead
            self._k("sed_transpose_shift", lib.sed_transpose_shift, g["hseq"].data_ptr() + 4 * d * Hd, 2 * Hd,
                    g["hprevT"].data_ptr() + 4 * d * Hd * Rp, Rp, R, Hd, t, 1 if d == 0 else -1, st)
            a_gi = g["dgiT"].data_ptr() + 4 * d * 3 * Hd * Rp
            a_gh = g["dghT"].data_ptr() + 4 * d * 3 * Hd * Rp
            ws = L.ptr(g["ws"]) if ks > 1 else None
            self._k("sed_gemm_nt", lib.sed_gemm_nt, dt, a_gi, Rp, L.ptr(g["mT"]), Rp, None,
                    L.ptr(G["gru.weight_ih_l0" + sfx]), Cl, 3 * Hd, Cl, R, ks, ws, st)
            self._k("sed_gemm_nt", lib.sed_gemm_nt, dt, a_gh, Rp, g["hprevT"].data_ptr() + 4 * d * Hd * Rp, Rp, None,
                    L.ptr(G["gru.weight_hh_l0" + sfx]), Hd, 3 * Hd, Hd, R, ks, ws, st)
            self._k("sed_row_sums", lib.sed_row_sums, a_gi, Rp, L.ptr(G["gru.bias_ih_l0" + sfx]), 3 * Hd, R, st)
            self._k("sed_row_sums", lib.sed_row_sums, a_gh, Rp, L.ptr(G["gru.bias_hh_l0" + sfx]), 3 * Hd, R, st)
            self._k("sed_transpose_shift", lib.sed_transpose_shift, L.ptr(P["gru.weight_ih_l0" + sfx]), Cl,
                    g["wihT"].data_ptr() + 4 * d * 3 * Hd, 6 * Hd, 3 * Hd, Cl, 3 * Hd, 0, st)
        if on_group_done is not None:
            on_group_done("gru")
        # dm = dgi . [W_ih ; W_ih_reverse]  (both directions in one K = 6*Hd product), then back through the mel mean
        self._k("sed_gemm_nt", lib.sed_gemm_nt, dt, L.ptr(g["dgi"]), 6 * Hd, L.ptr(g["wihT"]), 6 * Hd, None, L.ptr(g["dm"]),
                Cl, R, Cl, 6 * Hd, 1, None, st)
        self._k("sed_mel_mean_bwd", lib.sed_mel_mean_bwd, dt, L.ptr(g["dm"]), L.ptr(dfeat), R, p.w_out, Cl, pad32(Cl), st)
        self._tag = ""

    # ------------------------------------------------------------------------------------------
    def _bn_names(self, bi, j):
        pre = f"conv_blocks.{bi}.bn{j + 1}."
        return pre + "weight", pre + "bias", pre + "running_mean", pre + "running_var"

    def _wg_bufs(self, p: _Plan, ly: _Layer):
        """(dwpack, workspace) pointers of a weight-gradient launch: deferred reduction -> (NULL, the layer's own slab workspace)"""
        if p.wg_defer:
            return None, L.ptr(ly.wgrad_ws)
        return L.ptr(ly.dwpack), L.ptr(p.wgrad_ws)

    def _wg_done(self, p: _Plan, ly: _Layer, gw: torch.Tensor):
        """after a weight-gradient launch: remember the slabs it left for the batched reduction"""
        if p.wg_defer:
            p.wg_pending.append((ly.wgrad_ws.data_ptr(), 0, gw.data_ptr(), int(self.lib.sed_wgrad_last_slabs()), 9 * ly.cinp * ly.coutp,
                                 ly.cout, ly.cin, ly.cinp, ly.coutp))

    def _wg_flush(self, p: _Plan):
        """one launch reduces every pending layer's slabs into the torch-layout gradients (same summation order as the inline reduction)"""
        if not p.wg_pending:
            return
        key = tuple(p.wg_pending)
        ent = p.wg_tables.get(key)
        if ent is None:
            rows, blk = [], 0
            for r in p.wg_pending:
                rows.append(list(r) + [blk])
                blk += (r[4] + 63) // 64
            ent = p.wg_tables[key] = (torch.tensor(rows, dtype=torch.int64).to(p.layers[0][0].z.device), len(rows), blk)
        desc, n, blocks = ent
        self._k("sed_wgrad_reduce_batch", self.lib.sed_wgrad_reduce_batch, L.ptr(desc), n, blocks, _stream())
        p.wg_pending = []

    def _pack_weights(self, p: _Plan, P: Dict[str, torch.Tensor], training: bool) -> None:
        """Every conv layer's MFMA operand images in ONE launch: the forward operators and, for a train step, the
        data-gradient operators (the weights do not change between a step's forward and backward).  The descriptor table
        lives on the device and is rebuilt only when a parameter tensor moved."""
        ents = []
        for bi in range(len(self.cfg)):
            for j in range(2):
                if bi == 0 and j == 0 and not self.generic_first:
                    continue
                ly = p.layers[bi][j]
                w = P[f"conv_blocks.{bi}.conv{j + 1}.weight"]
                ents.append((w.data_ptr(), ly.wpack.data_ptr(), ly.cout, ly.cin, ly.coutp, ly.cinp, 0))
                if training:
                    ents.append((w.data_ptr(), ly.wpack_t.data_ptr(), ly.cout, ly.cin, ly.cinp, ly.coutp, 1))
        key = tuple(ents)
        cache = getattr(p, "pack_tables", None)
        if cache is None:
            cache = p.pack_tables = {}
        if key not in cache:
            rows, blk = [], 0
            for (wp, op, co, ci, pop, pip_, tf) in ents:
                rows.append([wp, op, co, ci, pop, pip_, tf, blk])
                blk += (pip_ * 9 * pop + 1023) // 1024
            cache[key] = (torch.tensor(rows, dtype=torch.int64).to(p.layers[0][0].z.device), len(rows), blk)
        desc, n, blocks = cache[key]
        if n:
            self._k("sed_pack_conv_weights_batch", self.lib.sed_pack_conv_weights_batch, self.dt_mm, L.ptr(desc), n, blocks, _stream())

    def forward(self, x: torch.Tensor, P: Dict[str, torch.Tensor], training: bool,
                feat_mean: Optional[torch.Tensor] = None, feat_std: Optional[torch.Tensor] = None,
                update_running_stats: bool = True) -> _Plan:
        """x: (B, 1, T, F) float32 cuda contiguous.  P: name -> fp32 cuda tensors (state_dict names).
        Leaves pre-interpolation logits in plan.pre (B, t, K)."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == self.cin0):
            raise ValueError(f"expected a float32 CUDA tensor of shape (B, {self.cin0}, T, F)")
        x = x.contiguous()
        B, _, T, F = x.shape
        p = self.plan(B, T, F, x.device)
        lib, dt, st = self.lib, self.dt, _stream()
        p.x_ref = x
        p.trained = training
        p.feat_mean, p.feat_std = feat_mean, feat_std
        prev = None
        if self.generic_first:
            if feat_mean is not None or feat_std is not None:
                raise ValueError("the z-score on load belongs to the Cin = 1 model path")
            self._k("sed_nchw_to_nhwc", lib.sed_nchw_to_nhwc, dt, L.ptr(x), L.ptr(p.x_nhwc), B, self.cin0, T, F,
                    p.x_nhwc.shape[3], st)
            prev = p.x_nhwc
        self._tag = ""
        self._pack_weights(p, P, training)
        for bi, (c, pool) in enumerate(self.cfg):
            for j in range(2):
                ly = p.layers[bi][j]
                w = P[f"conv_blocks.{bi}.conv{j + 1}.weight"]
                gname, bname, rmname, rvname = self._bn_names(bi, j)
                first = (bi == 0 and j == 0 and not self.generic_first)
                self._tag = f"fwd b{bi}c{j + 1} {ly.cin}->{ly.cout} H{ly.H} W{ly.W}"
                part = ly.part if training else None
                c1m = p.c1_mode and bi == 0
                if first and c1m:
                    pass        # z1 is never materialised: its consumers recompute it from x (BN1 statistics: Gram, below)
                elif first:
                    self._k("sed_conv3x3_c1_fwd", self.lib.sed_conv3x3_c1_fwd, dt, L.ptr(x), L.ptr(feat_mean), L.ptr(feat_std), L.ptr(w),
                                                   L.ptr(ly.z), L.ptr(part), B, ly.H, ly.W, ly.cout, ly.coutp, st)
                elif c1m and j == 1:
                    l1 = p.layers[bi][0]
                    self._k("sed_conv3x3_fwd_c1", self.lib.sed_conv3x3_fwd_c1, dt, L.EPI_STATS if training else L.EPI_STORE, L.ptr(x),
                            L.ptr(feat_mean), L.ptr(feat_std), L.ptr(P["conv_blocks.0.conv1.weight"]), L.ptr(l1.scale), L.ptr(l1.shift),
                            L.ptr(ly.wpack), L.ptr(ly.z), L.ptr(part), L.ptr(p.c1_mask) if (training and p.c1_mask is not None) else None, B, ly.H, ly.W,
                            ly.coutp, st)
                else:
                    if j == 0:
                        src, pro, ps, ph = prev, L.PRO_NONE, None, None
                    else:
                        l1 = p.layers[bi][0]
                        src, pro, ps, ph = l1.z, L.PRO_BNRELU, l1.scale, l1.shift
                    self._k("sed_conv3x3_fwd", self.lib.sed_conv3x3_fwd, self.dt_mm, pro, L.EPI_STATS if training else L.EPI_STORE, L.ptr(src),
                                                L.ptr(ps), L.ptr(ph), L.ptr(ly.wpack), L.ptr(ly.z), None, None, None,
                                                None, None, L.ptr(part), B, ly.H, ly.W, ly.cinp, ly.coutp, st)
                if training and first and c1m:
                    rm = P[rmname] if update_running_stats else None
                    rv = P[rvname] if update_running_stats else None
                    self._k("sed_conv3x3_c1_gram", self.lib.sed_conv3x3_c1_gram, L.ptr(x), L.ptr(feat_mean), L.ptr(feat_std),
                            L.ptr(p.c1_gram), B, ly.H, ly.W, st)
                    gram, ng, cnt = p.c1_gram, p.c1_gram.shape[0], float(B * ly.H * ly.W)
                    if self.bn_sync is not None:
                        gram, ng, cnt = self._sync_row(p.c1_gram, ng, 54, p.sync_gram), 1, cnt * self.bn_sync.world
                    if p.c1_tail and self.bn_sync is None:
                        self._k("sed_bn_train_finalize_c1", self.lib.sed_bn_train_finalize_c1_g, L.ptr(gram), ng,
                                cnt, L.ptr(w), L.ptr(P[gname]), L.ptr(P[bname]), L.ptr(rm), L.ptr(rv), BN_MOMENTUM, BN_EPS,
                                L.ptr(ly.scale), L.ptr(ly.shift), L.ptr(ly.mean), L.ptr(ly.invstd), ly.cout, ly.coutp, L.ptr(p.c1_gsum), st)
                        continue
                    self._k("sed_bn_train_finalize_c1", self.lib.sed_bn_train_finalize_c1, L.ptr(gram), ng,
                            cnt, L.ptr(w), L.ptr(P[gname]), L.ptr(P[bname]), L.ptr(rm), L.ptr(rv), BN_MOMENTUM, BN_EPS,
                            L.ptr(ly.scale), L.ptr(ly.shift), L.ptr(ly.mean), L.ptr(ly.invstd), ly.cout, ly.coutp, st)
                    continue
                if training:
                    rm = P[rmname] if update_running_stats else None
                    rv = P[rvname] if update_running_stats else None
                    part, npart, cnt = ly.part, ly.part.shape[0], float(B * ly.H * ly.W)
                    if self.bn_sync is not None:
                        part, npart, cnt = self._sync_row(ly.part, npart, 2 * ly.coutp, p.sync_row), 1, cnt * self.bn_sync.world
                    self._k("sed_bn_train_finalize", self.lib.sed_bn_train_finalize, L.ptr(part), npart, cnt,
                                                      L.ptr(P[gname]), L.ptr(P[bname]), L.ptr(rm), L.ptr(rv),
                                                      BN_MOMENTUM, BN_EPS, L.ptr(ly.scale), L.ptr(ly.shift),
                                                      L.ptr(ly.mean), L.ptr(ly.invstd), ly.cout, ly.coutp, st)
                else:
                    self._k("sed_bn_eval_coeffs", self.lib.sed_bn_eval_coeffs, L.ptr(P[gname]), L.ptr(P[bname]), L.ptr(P[rmname]),
                                                   L.ptr(P[rvname]), BN_EPS, L.ptr(ly.scale), L.ptr(ly.shift),
                                                   ly.cout, ly.coutp, st)
            l2 = p.layers[bi][1]
            self._tag = f"fwd b{bi}"
            if training and p.pool_fused[bi]:
                self._k("sed_bn_relu_pool_cnt_fwd", self.lib.sed_bn_relu_pool_cnt_fwd, dt, L.ptr(l2.z), L.ptr(l2.scale), L.ptr(l2.shift),
                        L.ptr(p.y[bi]), L.ptr(p.pool_cnt[bi]), B, l2.H, l2.W, l2.coutp, st)
            else:
                self._k("sed_bn_relu_pool_fwd", self.lib.sed_bn_relu_pool_fwd, dt, L.ptr(l2.z), L.ptr(l2.scale), L.ptr(l2.shift), L.ptr(p.y[bi]), B,
                                                 l2.H, l2.W, l2.coutp, pool, st)
            prev = p.y[bi]
        Cl = self.cfg[-1][0]
        self._tag = ""
        if self.head == "gru":
            self._gru_forward(p, P, prev, training)
            return p
        if self.head == "none":
            return p
        self._k("sed_head_fwd", self.lib.sed_head_fwd, dt, L.ptr(prev), L.ptr(P["event_fc.weight"]), L.ptr(P["event_fc.bias"]), L.ptr(p.m),
                                 L.ptr(p.pre), B, p.t_out, p.w_out, Cl, pad32(Cl), self.K, st)
        return p

    def interpolate(self, p: _Plan) -> torch.Tensor:
        out = torch.empty((p.B, p.t_out * self.ratio, self.K), dtype=torch.float32, device=p.pre.device)
        self._k("sed_interpolate", self.lib.sed_interpolate, L.ptr(p.pre), L.ptr(out), p.B, p.t_out, self.K, self.ratio, _stream())
        return out

    def loss_and_grad(self, p: _Plan, target: torch.Tensor, recall_factor: float, need_grad: bool = True,
                      grad_scale: float = 1.0) -> torch.Tensor:
        """WeightedBCE on the virtually interpolated logits; fills plan.dpre; returns plan.loss (1,)."""
        if not (target.is_cuda and target.dtype == torch.float32 and target.dim() == 3):
            raise ValueError("target must be a float32 CUDA tensor (B, T, K)")
        target = target.contiguous()
        self._k("sed_bce_fwd_bwd", self.lib.sed_bce_fwd_bwd, L.ptr(p.pre), L.ptr(target), L.ptr(p.loss),
                                         L.ptr(p.dpre) if need_grad else None, L.ptr(p.loss_partial), p.B, p.t_out,
                                         self.K, self.ratio, target.shape[1], float(recall_factor), float(grad_scale),
                                         _stream())
        return p.loss

    # ------------------------------------------------------------------------------------------
    def backward(self, p: _Plan, P: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor],
                 dlogits: Optional[torch.Tensor] = None, debug: Optional[dict] = None, on_group_done=None):
        """Backward of the last training-mode forward on plan p.  Gradient source: `dlogits`
        (B, t*ratio, K) w.r.t. the interpolated logits, or plan.dpre when None.  Writes fp32 gradients
        into G[name] (tensors shaped like the parameters; overwritten, not accumulated)."""
        if not p.trained:
            raise RuntimeError("backward() needs a training-mode forward (batch statistics)")
        lib, dt, st = self.lib, self.dt, _stream()
        B = p.B
        Cl = self.cfg[-1][0]
        if self.head == "none":
            src, ratio = None, 1
        elif dlogits is None:
            src, ratio = p.dpre, 1
        else:
            src, ratio = dlogits.contiguous(), self.ratio
        nb = len(self.cfg)
        if self.head == "none":
            pass                      # the caller filled plan.dy[-1] (gradient of the last pooled block output)
        elif self.head == "gru":
            self._gru_backward(p, P, G, src, ratio, p.dy[nb - 1], on_group_done)
        else:
            self._k("sed_head_bwd", self.lib.sed_head_bwd, dt, L.ptr(src), L.ptr(p.m), L.ptr(P["event_fc.weight"]),
                                     L.ptr(G["event_fc.weight"]), L.ptr(G["event_fc.bias"]), L.ptr(p.dy[nb - 1]),
                                     L.ptr(p.head_ws), B, p.t_out, p.w_out, Cl, pad32(Cl), self.K, ratio, st)
            if on_group_done is not None:
                on_group_done("event_fc")
        dzA, dzB = p.scratch
        if any(p.pool_fused):
            p.pool_flag.zero_()
        p.wg_pending = []

        def group_done(key):
            # data parallel: a group's gradients must be complete before its bucket's all-reduce is issued -- the pending reductions go out
            # per block there (four launches instead of seven); single process: one launch at the end of the backward
            if self.wg_flush_per_group:
                self._wg_flush(p)
            if on_group_done is not None:
                on_group_done(key)

        def snap(name, buf, ly):
            if debug is not None:
                debug[name] = buf[: B * ly.H * ly.W * ly.coutp].view(B, ly.H, ly.W, ly.coutp).float().clone()

        if debug is not None:
            debug[f"dy{nb - 1}"] = p.dy[nb - 1].float().clone()
        for bi in reversed(range(nb)):
            c, pool = self.cfg[bi]
            l1, l2 = p.layers[bi]
            H, W = l2.H, l2.W
            count = float(B * H * W)
            dtg = self._grad_dtype(B, H, W)
            self._tag = f"bwd b{bi}c2 {l2.cin}->{l2.cout} H{H} W{W}"
            g2n, b2n, _, _ = self._bn_names(bi, 1)
            g1n, b1n, _, _ = self._bn_names(bi, 0)
            # ---- pool + ReLU + BN2 backward -> dz2 -------------------------------------------------
            if p.pool_fused[bi]:
                # the statistics came out of the data gradient that produced dy[bi] (end of the previous iteration); the per-pixel
                # pass runs only if that kernel met a channel with gamma = 0 (device-side flag: the launch returns at once)
                nparts = p.pool_nparts[bi]
                self._k("sed_pool_relu_bwd_stats_if", self.lib.sed_pool_relu_bwd_stats_if, L.ptr(p.pool_flag[bi:]), dt, L.ptr(p.dy[bi]),
                        L.ptr(l2.z), L.ptr(l2.scale), L.ptr(l2.shift), L.ptr(l2.mean), L.ptr(l2.invstd), L.ptr(p.bwd_part), nparts,
                        B, H, W, l2.coutp, pool, st)
            else:
                nparts = lib.sed_pool_bwd_nparts(B, H, W, l2.coutp)
                self._k("sed_pool_relu_bwd_stats", self.lib.sed_pool_relu_bwd_stats, dt, L.ptr(p.dy[bi]), L.ptr(l2.z), L.ptr(l2.scale), L.ptr(l2.shift),
                                                    L.ptr(l2.mean), L.ptr(l2.invstd), L.ptr(p.bwd_part), B, H, W,
                                                    l2.coutp, pool, st)
            ca, cb, cc = l2.coef[0], l2.coef[1], l2.coef[2]
            sync = self.bn_sync
            gcount = count * (sync.world if sync is not None else 1)
            bpart = p.bwd_part
            if sync is not None:        # (dgamma / dbeta then come out as GLOBAL sums: pre-divided by world, the gradient
                bpart, nparts = self._sync_row(p.bwd_part, nparts, 2 * l2.coutp, p.sync_row), 1     # all-reduce sums them back)
            self._k("sed_bn_bwd_finalize", self.lib.sed_bn_bwd_finalize, L.ptr(bpart), nparts, gcount, L.ptr(P[g2n]), L.ptr(l2.mean),
                                            L.ptr(l2.invstd), L.ptr(G[g2n]), L.ptr(G[b2n]), L.ptr(ca), L.ptr(cb),
                                            L.ptr(cc), l2.cout, l2.coutp, st)
            if sync is not None:
                G[g2n].mul_(1.0 / sync.world)
                G[b2n].mul_(1.0 / sync.world)
            # ---- conv2 weight gradient; dz2 = BN2/ReLU/pool backward is produced on load inside the kernel
            #      (and written to dzA for the data-gradient call); its input a1 = relu(bn1(z1)) is
            #      recomputed on load as well -----------------------------------------------------------
            w2n = f"conv_blocks.{bi}.conv2.weight"
            c1m = p.c1_mode and bi == 0
            if c1m and debug is not None:
                raise RuntimeError("stage snapshots need conv1's output in memory: set SED_C1_MODE=0 (or use precision='fp32')")
            fused2 = p.bwd_fused[bi][1] and debug is None
            if fused2:
                # weight gradient AND gated data gradient from one dz2 tile in LDS: dz2 is never written, z1 is read once
                self._k("sed_conv3x3_bwd_fused", self.lib.sed_conv3x3_bwd_fused, dt, L.PRO_BNRELU, L.ptr(l1.z), L.ptr(l1.scale), L.ptr(l1.shift),
                        L.DZ_POOL, L.ptr(p.dy[bi]), L.ptr(l2.z), L.ptr(l2.scale), L.ptr(l2.shift), L.ptr(ca), L.ptr(cb), L.ptr(cc), pool,
                        L.ptr(l2.wpack_t), L.ptr(dzB), L.EPI_RELUBWD, L.ptr(l1.z), None, L.ptr(l1.scale), L.ptr(l1.shift), L.ptr(l1.mean),
                        L.ptr(l1.invstd), L.ptr(p.bwd_part), lib.sed_conv_nparts(B, H, W), None, *self._wg_bufs(p, l2),
                        B, H, W, l2.cinp, l2.coutp, L.ptr(G[w2n]), l2.cout, l2.cin, st)
            elif c1m and p.c1_bwd_fused and debug is None:
                # dW2 and the [A; sum g] partials of the gated data gradient from one dz2 tile in LDS: dz2 is never written
                x1a = (L.ptr(p.x_ref), L.ptr(p.feat_mean), L.ptr(p.feat_std), L.ptr(P["conv_blocks.0.conv1.weight"]))
                self._k("sed_conv3x3_bwd_fused_c1", self.lib.sed_conv3x3_bwd_fused_c1, dt, *x1a, L.ptr(l1.scale), L.ptr(l1.shift),
                        L.ptr(p.dy[bi]), L.ptr(l2.z), L.ptr(l2.scale), L.ptr(l2.shift), L.ptr(ca), L.ptr(cb), L.ptr(cc), pool,
                        L.ptr(l2.wpack_t), None if p.c1_gate_derived else L.ptr(p.c1_mask), L.ptr(p.c1_a10_part), *self._wg_bufs(p, l2),
                        B, H, W, l2.coutp, L.ptr(G[w2n]), l2.cout, l2.cin, st)
            elif c1m:
                x1a = (L.ptr(p.x_ref), L.ptr(p.feat_mean), L.ptr(p.feat_std), L.ptr(P["conv_blocks.0.conv1.weight"]))
                self._k("sed_conv3x3_wgrad_fused_c1", self.lib.sed_conv3x3_wgrad_fused_c1_u, dt, *x1a, L.ptr(l1.scale), L.ptr(l1.shift),
                        L.ptr(p.dy[bi]), L.ptr(l2.z), L.ptr(l2.scale), L.ptr(l2.shift), L.ptr(ca), L.ptr(cb), L.ptr(cc), pool,
                        L.ptr(dzA), *self._wg_bufs(p, l2), B, H, W, l2.coutp, L.ptr(G[w2n]), l2.cout, l2.cin, st)
            else:
                self._k("sed_conv3x3_wgrad_fused", self.lib.sed_conv3x3_wgrad_fused_u, dtg, L.PRO_BNRELU, L.ptr(l1.z),
                        L.ptr(l1.scale), L.ptr(l1.shift), L.DZ_POOL, L.ptr(p.dy[bi]), L.ptr(l2.z), L.ptr(l2.scale),
                        L.ptr(l2.shift), L.ptr(ca), L.ptr(cb), L.ptr(cc), pool, L.ptr(dzA), *self._wg_bufs(p, l2),
                        B, H, W, l2.cinp, l2.coutp, L.ptr(G[w2n]), l2.cout, l2.cin, st)
            self._wg_done(p, l2, G[w2n])
            snap(f"dz2_{bi}", dzA, l2)
            # (the reduction kernel of the weight gradient stores G[w2n] in torch layout itself; the data-gradient operator
            #  wpack_t was packed with the forward operators, sed_pack_conv_weights_batch)
            # ---- conv2: data gradient with fused ReLU mask + BN1 backward statistics ---------------
            nparts = lib.sed_conv_nparts(B, H, W)
            c1f = c1m and p.c1_dg_fused and debug is None
            if fused2 or (c1f and p.c1_bwd_fused):
                pass
            elif c1f:
                # g is never written: the kernel gates conv2^T(dz2) with the ReLU mask in registers and reduces it to
                # A = sum_px g (x) patch and sum g on the matrix pipe
                self._k("sed_conv3x3_dgrad_c1_stats", self.lib.sed_conv3x3_dgrad_c1_stats, dt, L.ptr(dzA), L.ptr(l2.wpack_t),
                        L.ptr(p.x_ref), L.ptr(p.feat_mean), L.ptr(p.feat_std), L.ptr(p.c1_mask), L.ptr(p.c1_a10_part), B, H, W, st)
            elif c1m:
                self._k("sed_conv3x3_dgrad_c1", self.lib.sed_conv3x3_dgrad_c1, dt, L.ptr(dzA), L.ptr(l2.wpack_t), L.ptr(dzB),
                        L.ptr(p.c1_mask), L.ptr(p.bwd_part), B, H, W, l2.coutp, st)
            else:
                self._k("sed_conv3x3_fwd", self.lib.sed_conv3x3_fwd, dtg, L.PRO_NONE, L.EPI_RELUBWD, L.ptr(dzA), None, None, L.ptr(l2.wpack_t),
                                            L.ptr(dzB), L.ptr(l1.z), L.ptr(l1.scale), L.ptr(l1.shift), L.ptr(l1.mean),
                                            L.ptr(l1.invstd), L.ptr(p.bwd_part), B, H, W, l2.coutp, l2.cinp, st)
            snap(f"g1_{bi}", dzB, l1)
            ca, cb, cc = l1.coef[0], l1.coef[1], l1.coef[2]
            c1_A = p.c1_A
            c1_tail = bool(c1f and p.c1_tail and sync is None and bi == 0)
            if c1_tail:
                # [A; sum g] partial rows -> BN1 backward coefficients -> dW1 in one launch, Gram statistics from the forward's finalize
                self._tag = f"bwd b{bi}c1 {l1.cin}->{l1.cout} H{H} W{W}"
                w1n_ = f"conv_blocks.{bi}.conv1.weight"
                self._k("sed_c1_bwd_tail", self.lib.sed_c1_bwd_tail, L.ptr(p.c1_a10_part), p.c1_a10_part.shape[0], L.ptr(p.c1_gsum), gcount,
                        L.ptr(P[w1n_]), L.ptr(P[g1n]), L.ptr(l1.mean), L.ptr(l1.invstd), L.ptr(G[g1n]), L.ptr(G[b1n]), L.ptr(ca), L.ptr(cb),
                        L.ptr(cc), L.ptr(p.c1_a10), L.ptr(l1.dwpack), l1.cout, l1.coutp, L.ptr(G[w1n_]), st)
            elif c1f:
                self._tag = f"bwd b{bi}c1 {l1.cin}->{l1.cout} H{H} W{W}"
                self._k("sed_sum_partials", self.lib.sed_sum_partials, L.ptr(p.c1_a10_part), p.c1_a10_part.shape[0], 10 * 32,
                        L.ptr(p.c1_a10), st)
                c1_A = p.c1_a10           # rows 0..8 = A, row 9 = sum g (read as a one-row statistics partial)
                a10 = p.c1_a10
                if sync is not None:      # BN1's backward statistics over the global batch; dW1's combine keeps the LOCAL A / Gram
                    p.sync_a10.copy_(p.c1_a10)
                    sync.all_reduce(p.sync_a10)
                    a10 = p.sync_a10
                self._k("sed_bn_bwd_finalize_c1", self.lib.sed_bn_bwd_finalize_c1, L.ptr(a10[9]), 1, gcount, L.ptr(a10),
                        L.ptr(P["conv_blocks.0.conv1.weight"]), L.ptr(P[g1n]), L.ptr(l1.mean), L.ptr(l1.invstd), L.ptr(G[g1n]),
                        L.ptr(G[b1n]), L.ptr(ca), L.ptr(cb), L.ptr(cc), l1.cout, l1.coutp, st)
            elif c1m:
                # BN1 backward needs sum g*z1 = w1 . A with A = the plain first-layer weight gradient of g1: that
                # kernel runs first, the coefficients come from sed_bn_bwd_finalize_c1
                if sync is not None:
                    raise RuntimeError("SyncBN needs the fused block-0 data gradient (SED_DGRAD_FUSED=1) or SED_C1_MODE=0")
                self._tag = f"bwd b{bi}c1 {l1.cin}->{l1.cout} H{H} W{W}"
                self._k("sed_conv3x3_c1_wgrad", self.lib.sed_conv3x3_c1_wgrad, dt, L.ptr(p.x_ref), L.ptr(p.feat_mean),
                        L.ptr(p.feat_std), L.ptr(dzB), L.ptr(p.c1_ws), B, H, W, l1.coutp, st)
                self._k("sed_sum_partials", self.lib.sed_sum_partials, L.ptr(p.c1_ws), p.c1_ws.shape[0], 9 * l1.coutp,
                        L.ptr(p.c1_A), st)
                self._k("sed_bn_bwd_finalize_c1", self.lib.sed_bn_bwd_finalize_c1, L.ptr(p.bwd_part), nparts, count, L.ptr(p.c1_A),
                        L.ptr(P["conv_blocks.0.conv1.weight"]), L.ptr(P[g1n]), L.ptr(l1.mean), L.ptr(l1.invstd), L.ptr(G[g1n]),
                        L.ptr(G[b1n]), L.ptr(ca), L.ptr(cb), L.ptr(cc), l1.cout, l1.coutp, st)
            else:
                bpart = p.bwd_part
                if sync is not None:
                    bpart, nparts = self._sync_row(p.bwd_part, nparts, 2 * l1.coutp, p.sync_row), 1
                self._k("sed_bn_bwd_finalize", self.lib.sed_bn_bwd_finalize, L.ptr(bpart), nparts, gcount, L.ptr(P[g1n]), L.ptr(l1.mean),
                                                L.ptr(l1.invstd), L.ptr(G[g1n]), L.ptr(G[b1n]), L.ptr(ca), L.ptr(cb),
                                                L.ptr(cc), l1.cout, l1.coutp, st)
            if sync is not None:
                G[g1n].mul_(1.0 / sync.world)
                G[b1n].mul_(1.0 / sync.world)
            w1n = f"conv_blocks.{bi}.conv1.weight"
            xin = p.y[bi - 1] if bi > 0 else getattr(p, "x_nhwc", None)
            dxout = p.dy[bi - 1] if bi > 0 else getattr(p, "dx", None)
            if bi == 0 and not self.generic_first:
                # first layer (Cin = 1): direct weight-gradient kernel with dz1 = BN1 backward computed on load
                # from (g1, z1); block 0 has no data gradient, so dz1 is never written to memory
                self._tag = f"bwd b{bi}c1 {l1.cin}->{l1.cout} H{H} W{W}"
                if debug is not None:
                    tmp = torch.empty_like(dzB[: B * H * W * l1.coutp])
                    self._k("sed_bn_bwd_apply", self.lib.sed_bn_bwd_apply, dt, L.ptr(dzB), L.ptr(l1.z), L.ptr(ca), L.ptr(cb),
                            L.ptr(cc), L.ptr(tmp), B * H * W, l1.coutp, st)
                    snap(f"dz1_{bi}", tmp, l1)
                # dW1 = ca*A + cb*(w1.G) + cc*sx: A = plain weight gradient of g1, G / sx = Gram statistics of the
                # input patches -- z1 is not read (csrc/sed_conv.hip: conv_c1_gram_kernel)
                if not c1m:          # (C1 mode: Gram statistics from the forward pass, A from the BN1 step above)
                    self._k("sed_conv3x3_c1_gram", self.lib.sed_conv3x3_c1_gram, L.ptr(p.x_ref), L.ptr(p.feat_mean),
                            L.ptr(p.feat_std), L.ptr(p.c1_gram), B, H, W, st)
                    self._k("sed_conv3x3_c1_wgrad", self.lib.sed_conv3x3_c1_wgrad, dt, L.ptr(p.x_ref), L.ptr(p.feat_mean),
                            L.ptr(p.feat_std), L.ptr(dzB), L.ptr(p.c1_ws), B, H, W, l1.coutp, st)
                    self._k("sed_sum_partials", self.lib.sed_sum_partials, L.ptr(p.c1_ws), p.c1_ws.shape[0], 9 * l1.coutp,
                            L.ptr(p.c1_A), st)
                if not c1_tail:      # (the tail kernel above has stored dW1 already)
                    self._k("sed_conv3x3_c1_wgrad_combine", self.lib.sed_conv3x3_c1_wgrad_combine_u, L.ptr(c1_A), L.ptr(p.c1_gram),
                            p.c1_gram.shape[0], L.ptr(P[w1n]), L.ptr(ca), L.ptr(cb), L.ptr(cc), L.ptr(l1.dwpack), l1.cout,
                            l1.coutp, L.ptr(G[w1n]), st)
            else:
                # conv1 weight gradient with dz1 = BN1 backward produced on load from (g1, z1); dz1 lands
                # in dzA (dz2 is dead by now) for the data-gradient call below
                self._tag = f"bwd b{bi}c1 {l1.cin}->{l1.cout} H{H} W{W}"
                if p.bwd_fused[bi][0] and debug is None:
                    pst = bi > 0 and p.pool_fused[bi - 1]
                    q2 = p.layers[bi - 1][1] if pst else None
                    self._k("sed_conv3x3_bwd_fused", self.lib.sed_conv3x3_bwd_fused, dt, L.PRO_NONE, L.ptr(xin), None, None, L.DZ_BN, L.ptr(dzB),
                            L.ptr(l1.z), None, None, L.ptr(ca), L.ptr(cb), L.ptr(cc), 1, L.ptr(l1.wpack_t), L.ptr(dxout),
                            L.EPI_POOLSTATS if pst else L.EPI_STORE, L.ptr(p.y[bi - 1]) if pst else None,
                            L.ptr(p.pool_cnt[bi - 1]) if pst else None, L.ptr(q2.scale) if pst else None, L.ptr(q2.shift) if pst else None,
                            L.ptr(q2.mean) if pst else None, L.ptr(q2.invstd) if pst else None, L.ptr(p.bwd_part) if pst else None,
                            p.pool_nparts[bi - 1] if pst else 0, L.ptr(p.pool_flag[bi - 1:]) if pst else None, *self._wg_bufs(p, l1),
                            B, H, W, l1.cinp, l1.coutp, L.ptr(G[w1n]), l1.cout, l1.cin, st)
                    self._wg_done(p, l1, G[w1n])
                    group_done(f"conv_blocks.{bi}")
                    continue
                self._k("sed_conv3x3_wgrad_fused", self.lib.sed_conv3x3_wgrad_fused_u, dtg, L.PRO_NONE, L.ptr(xin),
                        None, None, L.DZ_BN, L.ptr(dzB), L.ptr(l1.z), None, None, L.ptr(ca), L.ptr(cb), L.ptr(cc), 1,
                        L.ptr(dzA), *self._wg_bufs(p, l1), B, H, W, l1.cinp, l1.coutp, L.ptr(G[w1n]), l1.cout,
                        l1.cin, st)
                self._wg_done(p, l1, G[w1n])
                snap(f"dz1_{bi}", dzA, l1)
                if bi > 0 and p.pool_fused[bi - 1]:
                    q2 = p.layers[bi - 1][1]      # the block whose pooled output this gradient belongs to
                    self._k("sed_conv3x3_dgrad_poolstats", self.lib.sed_conv3x3_dgrad_poolstats, dt, L.ptr(dzA), L.ptr(l1.wpack_t),
                            L.ptr(dxout), L.ptr(p.y[bi - 1]), L.ptr(p.pool_cnt[bi - 1]), L.ptr(q2.scale), L.ptr(q2.shift), L.ptr(q2.mean),
                            L.ptr(q2.invstd), L.ptr(p.bwd_part), p.pool_nparts[bi - 1], L.ptr(p.pool_flag[bi - 1:]), B, H, W,
                            l1.coutp, l1.cinp, st)
                else:
                    self._k("sed_conv3x3_fwd", self.lib.sed_conv3x3_fwd, dtg, L.PRO_NONE, L.EPI_STORE, L.ptr(dzA), None, None,
                            L.ptr(l1.wpack_t), L.ptr(dxout), None, None, None, None, None, None, B, H, W, l1.coutp,
                            l1.cinp, st)
                if debug is not None and bi > 0:
                    debug[f"dy{bi - 1}"] = p.dy[bi - 1].float().clone()
            group_done(f"conv_blocks.{bi}")
        self._wg_flush(p)
        self._tag = ""

    # ------------------------------------------------------------------------------------------
    def adam_step_dev(self, flat_p, flat_g, flat_m, flat_v, flat_vmax, hyper, step_dev, grad_scale: float, lr_decay: float,
                      decay_every: int, betas=(0.9, 0.999), eps: float = 1e-8):
        """Adam-amsgrad with the step counter / learning rate / bias corrections in device memory (graph replay)."""
        self._k("sed_adam_amsgrad_step_dev", self.lib.sed_adam_amsgrad_step_dev, L.ptr(flat_p), L.ptr(flat_g), L.ptr(flat_m),
                L.ptr(flat_v), L.ptr(flat_vmax), flat_p.numel(), L.ptr(hyper), L.ptr(step_dev), float(betas[0]), float(betas[1]),
                float(eps), float(grad_scale), float(lr_decay), int(decay_every), _stream())

    def adam_step(self, flat_p, flat_g, flat_m, flat_v, flat_vmax, lr: float, step: int, grad_scale: float = 1.0,
                  betas=(0.9, 0.999), eps: float = 1e-8):
        self._k("sed_adam_amsgrad_step", self.lib.sed_adam_amsgrad_step, L.ptr(flat_p), L.ptr(flat_g), L.ptr(flat_m), L.ptr(flat_v),
                                               L.ptr(flat_vmax), flat_p.numel(), float(lr), float(betas[0]),
                                               float(betas[1]), float(eps), int(step), float(grad_scale), _stream())
