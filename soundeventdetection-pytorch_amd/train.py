"""Counterpart of /root/reference/train.py (train :77-131, eval :12-74) for the MI355X pipeline.

`train()` / `eval()` keep the reference signatures.  The step itself is the fused path
    forward (HIP) -> WeightedBCE fwd+bwd on the un-materialised x8 logits (HIP) -> backward (HIP)
    -> [RCCL all-reduce of the flat fp32 gradient buffer, bucketed per ConvBlock and overlapped with
       the rest of backward] -> fused Adam-amsgrad on the flat parameter buffer (HIP)
with the reference's semantics: Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=True)
(train.py:85), `lr *= 0.997` after every 200th iteration (train.py:108-110), checkpoint dict keys
{'iterations', 'model', 'optimizer'} (train.py:123-128).

Data parallelism (one process per GPU, torch.distributed backend "nccl" == RCCL over xGMI): each
rank runs the same step on its shard of the global batch; gradients are averaged with one
all-reduce per bucket issued as soon as the bucket's last gradient kernel has been enqueued, so
the collective of block i overlaps the backward kernels of blocks i-1..0.  BatchNorm statistics are
per rank (the torch DDP convention); optimizer state is replicated.
"""
from __future__ import annotations

import json
import os
from time import time
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib as L
from .utils.metric_utils import calculate_metrics, calculate_metrics_device, f_score  # noqa: F401

LR_DECAY_FREQ = 200      # train.py:80
LR_DECAY = 0.997         # train.py:110


# ----------------------------------------------------------------------------------------------
# flat parameter / gradient storage
# ----------------------------------------------------------------------------------------------
class FlatParams:
    """All trainable parameters of a model as views of ONE fp32 buffer (and their gradients as views
    of another), in nn.Module.parameters() order, each start padded to 4 floats.  The per-block
    slices [start, end) are the all-reduce buckets."""

    def __init__(self, model: torch.nn.Module, n_buckets: Optional[int] = None):
        """n_buckets: how many all-reduce buckets the per-group slices are merged into (None: $SED_DDP_BUCKETS, default 2;
        0: one bucket per top-level group, the round-1 layout kept for A/B runs; 1: a single flat all-reduce)."""
        named = list(model.named_parameters())
        if not named:
            raise ValueError("model has no parameters")
        dev = named[0][1].device
        self.names = [n for n, _ in named]
        self.offsets: Dict[str, int] = {}
        off = 0
        for n, p in named:
            self.offsets[n] = off
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        self.p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.P: Dict[str, torch.Tensor] = {}
        self.G: Dict[str, torch.Tensor] = {}
        for n, p in named:
            o = self.offsets[n]
            view = self.p[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view                      # the nn.Parameter now aliases the flat buffer
            self.P[n] = view
            self.G[n] = self.g[o:o + p.numel()].view(p.shape)
        self.model = model
        self.groups = self._make_buckets()
        if n_buckets is None:
            n_buckets = int(os.environ.get("SED_DDP_BUCKETS", "2"))
        self.buckets = self._merge_buckets(self.groups, n_buckets)

    def _make_buckets(self) -> List[tuple]:
        """(start, end) per top-level group in BACKWARD completion order: event_fc first, then
        conv_blocks.N-1 ... conv_blocks.0."""
        groups: Dict[str, List[int]] = {}
        order: List[str] = []
        for n in self.names:
            key = ".".join(n.split(".")[:2]) if n.startswith("conv_blocks.") else n.split(".")[0]
            if key not in groups:
                groups[key] = [self.offsets[n], 0]
                order.append(key)
            numel = int(np.prod(self.P[n].shape))
            groups[key][1] = self.offsets[n] + (numel + 3) // 4 * 4
        return [(k, groups[k][0], groups[k][1]) for k in reversed(order)]

    @staticmethod
    def _merge_buckets(groups, n_buckets: int, head_share: float = 0.85):
        """Merge the per-group slices (backward completion order, contiguous in the flat buffer) into at most `n_buckets`
        all-reduces.  The gradient buffer is 2.3 MB: every collective is latency-bound, so fewer is better; two
        keep the overlap -- the head bucket (late layers, >= 85 % of the elements) goes out while the first blocks' backward
        still runs, the small tail bucket is the only exposed one.  Each entry: (keys, start, end); a bucket is ready when
        the LAST of its keys is."""
        if n_buckets <= 0 or n_buckets >= len(groups):
            return [((k,), s, e) for (k, s, e) in groups]
        total = sum(e - s for _, s, e in groups)
        if n_buckets == 1:
            cuts = [len(groups)]
        else:
            acc, cut = 0, len(groups) - 1
            for i, (_, s, e) in enumerate(groups[:-1]):
                acc += e - s
                if acc >= head_share * total:
                    cut = i + 1
                    break
            cuts = [cut, len(groups)]
            # (more than two buckets: split the head evenly by group count)
            if n_buckets > 2 and cut > 1:
                step = max(1, cut // (n_buckets - 1))
                cuts = sorted(set(list(range(step, cut, step))[: n_buckets - 2] + [cut, len(groups)]))
        out, lo = [], 0
        for hi in cuts:
            part = groups[lo:hi]
            if part:
                out.append((tuple(k for k, _, _ in part), min(s for _, s, _ in part), max(e for _, _, e in part)))
            lo = hi
        return out

    def aliased(self) -> bool:
        """False once something (e.g. model.to()) replaced the parameter storages."""
        for n, p in self.model.named_parameters():
            if p.data_ptr() != self.P[n].data_ptr():
                return False
        return True

    def tensor_dict(self) -> Dict[str, torch.Tensor]:
        d = dict(self.P)
        d.update({n: b for n, b in self.model.named_buffers()})
        return d


class GradAllReducer:
    """Bucketed, overlapped gradient averaging over torch.distributed (RCCL on the GPU box, gloo
    in the CPU tests).  No-op for world_size 1.  `buckets`: [(keys, start, end)] from FlatParams (a bare (key, start, end)
    triple is accepted as a one-key bucket)."""

    def __init__(self, flat_g: torch.Tensor, buckets, group=None):
        import torch.distributed as dist
        self.dist = dist
        # SED_DDP_FORCE=1 (test hook): run the collectives on a world-size-1 group too, so that one GPU can execute the RCCL path
        # (librccl, async handles, stream ordering) -- tests/test_gpu_ddp.py
        force = os.environ.get("SED_DDP_FORCE", "0") == "1"
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.flat_g, self.group = flat_g, group
        self.buckets = [((k,) if isinstance(k, str) else tuple(k), s, e) for (k, s, e) in buckets]
        self._bucket_of = {k: i for i, (keys, _, _) in enumerate(self.buckets) for k in keys}
        self._waiting = [set(keys) for keys, _, _ in self.buckets]
        self.pending = []
        self.issued = []           # bucket indices in issue order of the current step (tests / traces)
        # bench.py --gpus N: set to a list to have finish() bracket its waits with two events on the current stream -- the time the
        # compute stream spends blocked on the collectives after the backward's last kernel = the EXPOSED communication of the step
        self.exposed_events = None

    def bucket_ready(self, key: str):
        """Call right after the kernels producing group `key` have been enqueued; the all-reduce of the bucket the group
        belongs to is issued (async) once all of its groups are ready."""
        if key not in self._bucket_of:
            raise KeyError(key)
        if not self.enabled:
            return
        i = self._bucket_of[key]
        self._waiting[i].discard(key)
        if not self._waiting[i]:
            _, s, e = self.buckets[i]
            self.issued.append(i)
            self.pending.append(self.dist.all_reduce(self.flat_g[s:e], op=self.dist.ReduceOp.SUM,
                                                     group=self.group, async_op=True))

    def finish(self) -> float:
        """Wait for every bucket; returns the factor the optimizer must apply (1/world)."""
        if self.enabled:
            for i, w in enumerate(self._waiting):      # a group nobody reported (model without that layer type): flush
                if w and i not in self.issued:
                    _, s, e = self.buckets[i]
                    self.issued.append(i)
                    self.pending.append(self.dist.all_reduce(self.flat_g[s:e], op=self.dist.ReduceOp.SUM,
                                                             group=self.group, async_op=True))
        ev = None
        if self.exposed_events is not None and self.pending:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self.pending:
            w.wait()
        if ev is not None:
            ev[1].record()
            self.exposed_events.append(ev)
        self.pending = []
        self.issued = []
        self._waiting = [set(keys) for keys, _, _ in self.buckets]
        return 1.0 / self.world


class BnSync:
    """SyncBN plumbing handed to the engine: in-place SUM all-reduce of a small fp32 row, ordered after the kernels already
    enqueued on the current stream (torch.distributed's synchronous collectives have exactly that stream semantics)."""

    def __init__(self, dist, group, world):
        self.dist, self.group, self.world = dist, group, int(world)
        self.calls = 0             # collectives issued so far (bench.py reports them per step)

    def all_reduce(self, t: torch.Tensor):
        self.calls += 1
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)


def seed_all_ranks(seed: Optional[int] = None, group=None) -> Optional[int]:
    """Identical host RNG state (random, numpy, torch) on every data-parallel rank.  The dataset classes shuffle with the
    global RNGs like the reference's do (spectograms_dataset.py:53,176-185; waveform_dataset.py split/shuffle): ranks that
    draw different train/val splits or start-index permutations would train on each other's validation files and the
    rank-sharded index ranges would overlap.  With seed=None rank 0 draws one and broadcasts it.  Single process: seeds
    only if a seed is given."""
    import random
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor([seed if seed is not None else random.getrandbits(31)], dtype=torch.int64, device=dev)
        dist.broadcast(t, src=0, group=group)
        seed = int(t.item())
    if seed is not None:
        random.seed(seed)
        np.random.seed(seed % (2 ** 32))
        torch.manual_seed(seed)
    return seed


def reseed_rank(seed: Optional[int], rank: int) -> None:
    """Rank-specific host RNG streams AFTER the rank-identical draws (train/val split, start-index permutation) are done:
    the datasets draw their augmentation decisions (mix-in partners, noise, gains) from the global numpy RNG in
    __getitem__ / device_batch, and with identical streams every rank would make the same draws at every step -- the
    augmentation diversity of the global batch would shrink by the world size against the reference's single process
    (spectograms_dataset.py:71-73,112-135).  Parameters are broadcast from rank 0 in FusedTrainer anyway."""
    import random
    if seed is None:
        return
    s = int(seed) + 1000003 * (int(rank) + 1)
    random.seed(s)
    np.random.seed(s % (2 ** 32))
    torch.manual_seed(s)


def sharded_batch_indices(n: int, batch_size: int, rank: int = 0, world_size: int = 1):
    """Index lists of one epoch for rank `rank`: idx = step*B_global + rank*B_local + i of the dataset's own order (SURVEY 8e;
    the reference builds its DataLoader without shuffle and without drop_last, main.py:125).  One process keeps the short tail
    batch like DataLoader does.  With world_size > 1 every rank must hold the same number of samples per step (the 1/world
    gradient average and SyncBN's `count * world` assume it), so the ragged tail of the LAST global batch is filled by wrapping
    to the start of the table -- the same rule on every rank; no sample is left out of an epoch and len() is
    ceil(n / B_global) at every world size (a few samples at the head are seen twice instead).
    TRAIN LOADERS ONLY: a metric / evaluation loop over a wrapped epoch would count the head samples twice -- the validation samplers
    of the datasets (get_validation_sampler) walk whole files on every rank and do not come through here; the waveform path's
    WaveformBatchLoader drops its ragged tail instead (multiples of 8 frames per rank)."""
    B, g = int(batch_size), int(batch_size) * int(world_size)
    if n <= 0:
        return
    if world_size == 1:
        for base in range(0, n, B):
            yield list(range(base, min(base + B, n)))
        return
    for base in range(0, n, g):
        lo = base + rank * B
        yield [(lo + i) % n for i in range(B)]


class ShardedBatchLoader:
    """DataLoader stand-in for map-style datasets under data parallel (sharded_batch_indices above)."""

    def __init__(self, dataset, batch_size: int, rank: int = 0, world_size: int = 1):
        self.dataset, self.batch_size, self.rank, self.world_size = dataset, int(batch_size), int(rank), int(world_size)

    def __len__(self):
        g = self.batch_size * self.world_size
        return (len(self.dataset) + g - 1) // g

    def indices(self):
        return sharded_batch_indices(len(self.dataset), self.batch_size, self.rank, self.world_size)

    def __iter__(self):
        for idx in self.indices():
            items = [self.dataset[i] for i in idx]
            yield tuple(torch.stack([torch.as_tensor(it[k]) for it in items]) for k in range(len(items[0])))


# ----------------------------------------------------------------------------------------------
# the fused optimizer + step
# ----------------------------------------------------------------------------------------------
class FusedTrainer:
    """Owns the flat buffers, the Adam-amsgrad state and the step counter for one model."""

    def __init__(self, model, lr: float, recall_factor: float = 5.0, betas=(0.9, 0.999), eps: float = 1e-8,
                 group=None, graph: bool = False, sync_bn: bool = False, n_buckets: Optional[int] = None):
        """graph=True (single process): after two eager steps per input shape the whole step -- forward, loss, backward,
        Adam-amsgrad with its step counter, learning rate and bias corrections in device memory -- is captured into a HIP
        graph and replayed.  For the reference's own small shapes (T = 30 crops, batch 4: ~90 launches of a few microseconds
        each) the eager step is bound by launch overhead, not by the GPU."""
        self.model = model
        self.engine = model.engine
        self.use_graph = bool(graph)
        self._graphs = {}           # input-shape key -> (graph, static x, static y, loss buffer)
        self._eager_seen = {}
        if not next(model.parameters()).is_cuda:
            raise RuntimeError("FusedTrainer needs the model on the GPU (model.to('cuda')); there is no CPU path")
        self.flat = FlatParams(model, n_buckets)
        self.m = torch.zeros_like(self.flat.p)
        self.v = torch.zeros_like(self.flat.p)
        self.vmax = torch.zeros_like(self.flat.p)
        self.lr = float(lr)
        self.betas, self.eps = betas, eps
        self.recall_factor = float(recall_factor)
        self.step_count = 0
        self.reducer = GradAllReducer(self.flat.g, self.flat.buckets, group)
        self.sync_bn = bool(sync_bn) and self.reducer.enabled
        if self.reducer.enabled and hasattr(self.engine, "wg_flush_per_group"):
            self.engine.wg_flush_per_group = True      # weight gradients complete before their bucket's all-reduce goes out
        if self.reducer.enabled:
            # replicas start from rank 0's parameters and BatchNorm buffers (the DDP constructor's broadcast): identical
            # seeds are not something to rely on
            self.reducer.dist.broadcast(self.flat.p, src=0, group=group)
            for _, b in model.named_buffers():
                if b.is_floating_point():
                    self.reducer.dist.broadcast(b, src=0, group=group)
        if self.sync_bn:
            if not hasattr(self.engine, "bn_sync"):
                raise RuntimeError("sync_bn is implemented for the spectrogram models (Cnn_AvgPooling / Crnn_AvgPooling)")
            self.engine.bn_sync = BnSync(self.reducer.dist, group, self.reducer.world)
        if self.use_graph:
            if self.reducer.enabled:
                raise RuntimeError("graph=True is the single-process path (the gradient collectives are not captured)")
            if not hasattr(self.engine, "adam_step_dev"):
                raise RuntimeError("graph=True needs an engine with a device-scalar optimizer step (Cnn_AvgPooling / Crnn_AvgPooling)")
            dev = self.flat.p.device
            self.hyper = torch.tensor([self.lr, 0.0, 0.0], dtype=torch.float32, device=dev)
            self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)

    def _check_alias(self):
        if not self.flat.aliased():
            raise RuntimeError("model parameters were re-allocated after FusedTrainer was built "
                               "(e.g. model.to()); build the trainer after moving the model")

    def forward_backward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """One forward + loss + backward; gradients land in flat.g; returns the device loss (1,)."""
        self._check_alias()
        eng = self.engine
        P = self.flat.tensor_dict()
        self.model.train()
        plan = eng.forward(x, P, training=True)
        self.model._nbt_pending += 1
        self.model._fwd_serial += 1
        loss = eng.loss_and_grad(plan, y, self.recall_factor)
        eng.backward(plan, P, self.flat.G, on_group_done=self.reducer.bucket_ready)
        return loss

    def optimizer_step(self):
        scale = self.reducer.finish()
        self.step_count += 1
        self.engine.adam_step(self.flat.p, self.flat.g, self.m, self.v, self.vmax, self.lr, self.step_count,
                              grad_scale=scale, betas=self.betas, eps=self.eps)
        if self.step_count % LR_DECAY_FREQ == 0:       # train.py:108-110 (after that iteration's step)
            self.lr *= LR_DECAY

    def _step_dev(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """forward + loss + backward + optimizer step with device-resident scalars: kernel launches only (capturable)."""
        loss = self.forward_backward(x, y)
        self.engine.adam_step_dev(self.flat.p, self.flat.g, self.m, self.v, self.vmax, self.hyper, self.step_dev, 1.0,
                                  LR_DECAY, LR_DECAY_FREQ, betas=self.betas, eps=self.eps)
        return loss

    def _host_mirror(self):
        self.step_count += 1
        if self.step_count % LR_DECAY_FREQ == 0:
            self.lr *= LR_DECAY

    def train_step(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """Returns the device loss buffer (1,) of this shape's plan: valid until the next step (clone to keep)."""
        if not self.use_graph:
            loss = self.forward_backward(x, y)
            self.optimizer_step()
            return loss
        key = (tuple(x.shape), tuple(y.shape))
        ent = self._graphs.get(key)
        if ent is None:
            seen = self._eager_seen.get(key, 0)
            if seen < 2:            # eager: allocates the plan, the descriptor tables, sets kernel attributes
                self._eager_seen[key] = seen + 1
                loss = self._step_dev(x, y)
                self._host_mirror()
                return loss
            xs, ys = torch.empty_like(x), torch.empty_like(y)
            xs.copy_(x); ys.copy_(y)
            nbt, ser = self.model._nbt_pending, self.model._fwd_serial
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss = self._step_dev(xs, ys)
            self.model._nbt_pending, self.model._fwd_serial = nbt, ser     # (capture does not execute the step)
            ent = self._graphs[key] = (g, xs, ys, loss)
        else:
            ent[1].copy_(x); ent[2].copy_(y)
        self._check_alias()
        self.model.train()
        self.model._nbt_pending += 1
        self.model._fwd_serial += 1
        ent[0].replay()
        self._host_mirror()
        return ent[3]

    def _sync_host_scalars(self):
        """graph mode keeps the authoritative step counter / learning rate on the device."""
        if self.use_graph and self._graphs:
            self.step_count = int(self.step_dev.item())
            self.lr = float(self.hyper[0].item())

    def state_dict(self):
        """torch.optim.Adam(amsgrad=True).state_dict() layout (what the reference saves under checkpoint['optimizer'],
        train.py:123-126): {'state': {i: {'step', 'exp_avg', 'exp_avg_sq', 'max_exp_avg_sq'}}, 'param_groups': [...]}, the
        parameters numbered in model.parameters() order -- loadable into torch.optim.Adam and back into this trainer."""
        self._sync_host_scalars()
        state = {}
        for i, n in enumerate(self.flat.names):
            o, shp = self.flat.offsets[n], self.flat.P[n].shape
            k = self.flat.P[n].numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.m[o:o + k].view(shp).clone(),
                        "exp_avg_sq": self.v[o:o + k].view(shp).clone(),
                        "max_exp_avg_sq": self.vmax[o:o + k].view(shp).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": True,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(self.flat.names)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Resume from state_dict() output or from a torch.optim.Adam(amsgrad=True) state_dict of the same model."""
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.flat.names):
            raise ValueError("optimizer state does not match this model's parameter list")
        if not groups[0].get("amsgrad", False):
            raise ValueError("the fused optimizer is Adam with amsgrad=True (train.py:85)")
        self.lr = float(groups[0]["lr"])
        self.betas, self.eps = tuple(groups[0]["betas"]), float(groups[0]["eps"])
        step = 0
        for i, n in enumerate(self.flat.names):
            o, k = self.flat.offsets[n], self.flat.P[n].numel()
            st = sd["state"].get(i)
            if st is None:
                self.m[o:o + k].zero_(); self.v[o:o + k].zero_(); self.vmax[o:o + k].zero_()
                continue
            self.m[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            self.vmax[o:o + k].copy_(st["max_exp_avg_sq"].reshape(-1))
            step = max(step, int(float(st["step"])))
        self.step_count = step
        if self.use_graph:
            self.hyper[0] = self.lr
            self.step_dev.fill_(step)


class FusedAdamAmsgrad:
    """torch.optim-like facade over the fused kernel for code that calls loss.backward() itself
    (gradients in p.grad): zero_grad() / step()."""

    def __init__(self, model, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.flat = FlatParams(model)
        self.model = model
        self.m = torch.zeros_like(self.flat.p)
        self.v = torch.zeros_like(self.flat.p)
        self.vmax = torch.zeros_like(self.flat.p)
        self.param_groups = [{"lr": float(lr)}]
        self.betas, self.eps, self.step_count = betas, eps, 0

    def zero_grad(self):
        for p in self.model.parameters():
            p.grad = None

    def step(self):
        self.flat.g.zero_()
        for n, p in self.model.named_parameters():
            if p.grad is not None:
                self.flat.G[n].copy_(p.grad)
        self.step_count += 1
        L.check(L.lib().sed_adam_amsgrad_step(L.ptr(self.flat.p), L.ptr(self.flat.g), L.ptr(self.m), L.ptr(self.v),
                                              L.ptr(self.vmax), self.flat.numel, float(self.param_groups[0]["lr"]),
                                              float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                              self.step_count, 1.0, torch.cuda.current_stream().cuda_stream),
                "adam_amsgrad_step")


# ----------------------------------------------------------------------------------------------
# reference-signature loops
# ----------------------------------------------------------------------------------------------
def eval(model, dataloader, criterion, outputs_dir, iteration, device, limit_val_samples=None):
    """train.py:12-74 minus the matplotlib figures: whole recordings, batch 1, running-stat BN,
    sigmoid, 21-threshold metrics.  Returns (losses, recall_sets, precision_sets, APs)."""
    losses, recal_sets, precision_sets, APs = [], [], [], []
    val_sampler = dataloader.dataset.get_validation_sampler(max_validate_num=limit_val_samples)
    for idx, (inp, target, file_name) in enumerate(val_sampler):
        model.eval()
        with torch.no_grad():
            output = model(inp.to(device).float())
        loss = criterion(output, target.to(device).float())
        output = output[0] if inp.dim() == 4 else output
        target = target[0] if inp.dim() == 4 else target.reshape(-1, 1)
        # sigmoid + the 21-threshold counting stay on the device (sed_metric_counts)
        recal_vals, precision_vals, AP = calculate_metrics_device(output, target.to(device).float(), raw_logits=True)
        losses.append(loss.item())
        recal_sets.append(recal_vals)
        precision_sets.append(precision_vals)
        APs.append(AP)
    return losses, recal_sets, precision_sets, APs


def summarize_validation(val_losses, recal_sets, precision_sets, APs):
    """ProgressPlotter.report_validation_metrics (utils/common.py:46-56): F-scores of the
    validation-AVERAGED precision/recall curves, including the swapped-argument call convention."""
    r = np.mean(recal_sets, axis=0)
    p = np.mean(precision_sets, axis=0)
    return {"val_loss": float(np.mean(val_losses)), "AP": float(np.mean(APs)),
            "max_f1": float(np.max(f_score(p, r, precision_importance_factor=1))),
            "max_f5": float(np.max(f_score(p, r, precision_importance_factor=5)))}


def train(model, data_loader, criterion, num_steps, lr, log_freq, outputs_dir, device):
    """train.py:77-131.  `criterion` must be this package's WeightedBCE(multi_frame=True) (its
    recall_factor feeds the fused loss kernel)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("the MI355X training path needs device='cuda' (there is no CPU path)")
    print("Training:")
    print("\t- Using device: ", device)
    os.makedirs(os.path.join(outputs_dir, "checkpoints"), exist_ok=True)
    model.to(dev)
    # the fused step computes WeightedBCE (utils/common.py:11-30) itself: honour exactly that criterion, refuse anything else
    from .utils.common import WeightedBCE
    if not isinstance(criterion, WeightedBCE):
        raise TypeError("train() runs the fused WeightedBCE loss kernel: pass this package's WeightedBCE "
                        f"(got {type(criterion).__name__}); other criteria would be silently ignored")
    wants_multi = hasattr(model, "conv_blocks")       # spectrogram models: frame-wise targets; M5: one label per frame
    if bool(criterion.multi_frame) != wants_multi:
        raise ValueError(f"{type(model).__name__} trains with WeightedBCE(multi_frame={wants_multi}) (main.py:44,71)")
    trainer = FusedTrainer(model, lr, recall_factor=criterion.recall_factor,
                           sync_bn=os.environ.get("SED_SYNC_BN", "0") == "1")
    rank0 = (not trainer.reducer.enabled) or trainer.reducer.dist.get_rank() == 0
    log_path = os.path.join(outputs_dir, "progress.jsonl")
    losses: List[float] = []
    iterations, epoch = 0, 0
    t0 = time()
    world = trainer.reducer.world
    while iterations < num_steps:
        seen = 0
        for (batch_features, event_labels) in data_loader:
            seen += 1
            loss = trainer.train_step(batch_features.to(dev, non_blocking=True).float(),
                                      event_labels.to(dev, non_blocking=True).float())
            losses.append(loss.clone())              # device scalars (the step's loss buffer is reused): no host sync
            iterations += 1
            if iterations % log_freq == 0:
                host_losses = [float(l) for l in torch.stack([l.reshape(()) for l in losses]).cpu()]
                losses = []
                im_sec = iterations * data_loader.batch_size * world / (time() - t0)      # whole job, all ranks
                rec = {"epoch": epoch, "step": iterations, "train_loss": float(np.mean(host_losses)),
                       "im_sec": im_sec, "lr": trainer.lr}
                if hasattr(data_loader.dataset, "get_validation_sampler"):
                    rec.update(summarize_validation(*eval(model, data_loader, criterion, outputs_dir,
                                                          iteration=iterations, device=dev, limit_val_samples=3)))
                if rank0:
                    print(f"epoch: {epoch}, step: {iterations}, loss: {host_losses[-1]:.2f}, "
                          f"im/sec: {im_sec:.1f}, lr: {trainer.lr:.8f}")
                    with open(log_path, "a") as f:
                        f.write(json.dumps(rec) + "\n")
                    torch.save({"iterations": iterations, "model": model.state_dict(),
                                "optimizer": trainer.state_dict()},
                               os.path.join(outputs_dir, "checkpoints", f"iteration_{iterations}.pth"))
            if iterations == num_steps:
                break
        if seen == 0:
            raise RuntimeError("the data loader produced no batch in a whole epoch (dataset smaller than "
                               "batch_size x world_size?): training cannot make progress")
        epoch += 1
    return trainer
