"""Child process of tests/test_gpu_ddp.py: one data-parallel rank of the fused train step on cuda:0.

    python tests/ddp_gpu_worker.py RANK WORLD PORT MODE OUT.pt

Several ranks share the one GPU of the test box, so the process group uses gloo (CUDA tensors are staged through the
host); the code path above the backend -- FlatParams buckets, GradAllReducer, BnSync, the parameter broadcast, the
engine's SyncBN hooks -- is the one the RCCL ranks run.  WORLD = 1 runs the same step without a process group (the
single-process reference the parent compares with).

MODE: same   every rank gets the full batch             -> parameters after 3 steps == single process, bit for bit
      shard  rank r gets clips [r*B/W, (r+1)*B/W)       -> averaged flat gradient after one forward/backward
      sync   shard + SyncBN                             -> logits / gradient of the single-process full batch
      samesync  every rank gets the full batch + SyncBN -> the global sums are exactly WORLD x the local ones: gradient of
                                                            the single-process run up to the last bit of the power-of-two scaling
      solo:r single process on shard r (WORLD must be 1)
"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = [(32, 2), (64, 2), (128, 2), (128, 1)]
B, T = 4, 64


def batch():
    g = torch.Generator().manual_seed(123)
    x = torch.randn(B, 1, T, 64, generator=g)
    y = (torch.rand(B, T, 1, generator=g) > 0.75).float()
    return x, y


def main():
    rank, world, port, mode, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import torch.distributed as dist
    backend = os.environ.get("SED_TEST_BACKEND", "gloo")     # "nccl" = RCCL: one rank per GPU, so world 1 on the test box
    grouped = world > 1 or backend == "nccl"
    if grouped:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group(backend, rank=rank, world_size=world)
    sed = importlib.import_module("soundeventdetection-pytorch_amd")
    torch.cuda.set_device(0)
    torch.manual_seed(0 if rank == 0 else 100 + rank)      # replicas must NOT rely on equal seeds: rank 0's weights win
    model = sed.Cnn_AvgPooling(1, CFG, precision=os.environ.get("SED_TEST_PRECISION", "fp32")).cuda()
    x, y = batch()
    if mode.startswith("solo:"):
        r, w = int(mode[5:]), 2
        x, y = x[r * B // w:(r + 1) * B // w], y[r * B // w:(r + 1) * B // w]
    elif mode in ("shard", "sync") and world > 1:
        x, y = x[rank * B // world:(rank + 1) * B // world], y[rank * B // world:(rank + 1) * B // world]
    x, y = x.cuda().contiguous(), y.cuda().contiguous()
    tr = sed.FusedTrainer(model, lr=1e-3, recall_factor=5.0, sync_bn=(mode in ("sync", "samesync")))
    if world == 1 and mode == "samesync" and not grouped:
        class _One:          # single-process twin of the SyncBN path: same row reduction (fp32 row sums), the all-reduce a no-op
            world = 1

            def all_reduce(self, t):
                return t
        model.engine.bn_sync = _One()
    res = {"buckets": [list(k) for k, _, _ in tr.flat.buckets]}
    if mode == "same":
        losses = []
        for _ in range(3):
            losses.append(float(tr.train_step(x, y).item()))
        res.update(losses=losses, p=tr.flat.p.cpu(), sd={k: v.cpu() for k, v in model.state_dict().items()})
    else:
        loss = tr.forward_backward(x, y)
        plan = next(iter(model.engine._plans.values()))
        logits = model.engine.interpolate(plan).cpu()
        issued = list(tr.reducer.issued)
        scale = tr.reducer.finish()
        res.update(loss=float(loss.item()), g=(tr.flat.g * scale).cpu(), logits=logits, issued=issued,
                   sd={k: v.cpu() for k, v in model.state_dict().items()})
    torch.cuda.synchronize()
    if rank == 0 or mode in ("sync", "shard", "samesync"):
        torch.save(res, out if world == 1 else f"{out}.r{rank}")
    if grouped:
        res_backend = dist.get_backend()
        dist.barrier()
        dist.destroy_process_group()
        print("process group backend:", res_backend, "world", world, "reducer enabled:", tr.reducer.enabled,
              "bn_sync:", type(getattr(model.engine, "bn_sync", None)).__name__)


if __name__ == "__main__":
    main()
