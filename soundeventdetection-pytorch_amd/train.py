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

    def __init__(self, model: torch.nn.Module):
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
        self.buckets = self._make_buckets()

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
    in the CPU tests).  No-op for world_size 1."""

    def __init__(self, flat_g: torch.Tensor, buckets, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.flat_g, self.buckets, self.group = flat_g, buckets, group
        self.pending = []

    def bucket_ready(self, key: str):
        """Call right after the kernels producing bucket `key` have been enqueued."""
        if not self.enabled:
            return
        for (k, s, e) in self.buckets:
            if k == key:
                self.pending.append(self.dist.all_reduce(self.flat_g[s:e], op=self.dist.ReduceOp.SUM,
                                                         group=self.group, async_op=True))
                return
        raise KeyError(key)

    def finish(self) -> float:
        """Wait for every bucket; returns the factor the optimizer must apply (1/world)."""
        for w in self.pending:
            w.wait()
        self.pending = []
        return 1.0 / self.world


# ----------------------------------------------------------------------------------------------
# the fused optimizer + step
# ----------------------------------------------------------------------------------------------
class FusedTrainer:
    """Owns the flat buffers, the Adam-amsgrad state and the step counter for one model."""

    def __init__(self, model, lr: float, recall_factor: float = 5.0, betas=(0.9, 0.999), eps: float = 1e-8,
                 group=None, graph: bool = False):
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
        self.flat = FlatParams(model)
        self.m = torch.zeros_like(self.flat.p)
        self.v = torch.zeros_like(self.flat.p)
        self.vmax = torch.zeros_like(self.flat.p)
        self.lr = float(lr)
        self.betas, self.eps = betas, eps
        self.recall_factor = float(recall_factor)
        self.step_count = 0
        self.reducer = GradAllReducer(self.flat.g, self.flat.buckets, group)
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

    def state_dict(self):
        return {"step": self.step_count, "lr": self.lr, "exp_avg": self.m, "exp_avg_sq": self.v,
                "max_exp_avg_sq": self.vmax, "param_names": self.flat.names, "offsets": self.flat.offsets,
                "betas": self.betas, "eps": self.eps, "amsgrad": True}


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
    trainer = FusedTrainer(model, lr, recall_factor=getattr(criterion, "recall_factor", 5.0))
    rank0 = (not trainer.reducer.enabled) or trainer.reducer.dist.get_rank() == 0
    log_path = os.path.join(outputs_dir, "progress.jsonl")
    losses: List[float] = []
    iterations, epoch = 0, 0
    t0 = time()
    while iterations < num_steps:
        for (batch_features, event_labels) in data_loader:
            loss = trainer.train_step(batch_features.to(dev, non_blocking=True).float(),
                                      event_labels.to(dev, non_blocking=True).float())
            losses.append(loss.clone())              # device scalars (the step's loss buffer is reused): no host sync
            iterations += 1
            if iterations % log_freq == 0:
                host_losses = [float(l) for l in torch.stack([l.reshape(()) for l in losses]).cpu()]
                losses = []
                im_sec = iterations * data_loader.batch_size / (time() - t0)
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
        epoch += 1
    return trainer
