"""world_size-2 data-parallel plumbing on CPU (gloo): flat gradient buckets in backward completion
order, asynchronous per-bucket all-reduce, 1/world scaling -- the same code path the GPU ranks run
with backend "nccl" (RCCL)."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

TINY_CFG = [(4, 2), (8, 2), (8, 2), (8, 1)]
MAIN_CFG = [(32, 2), (64, 2), (128, 2), (128, 1)]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sed = importlib.import_module("soundeventdetection-pytorch_amd")
        tr = importlib.import_module("soundeventdetection-pytorch_amd.train")
        torch.manual_seed(0)
        model = sed.Cnn_AvgPooling(1, MAIN_CFG)          # CPU module: only its parameter layout is used
        flat = tr.FlatParams(model)
        # parameters alias the flat buffer
        assert flat.aliased()
        w = model.conv_blocks[1].conv2.weight
        flat.p[flat.offsets["conv_blocks.1.conv2.weight"]] = 123.0
        assert w.data.flatten()[0].item() == 123.0
        keys = [k for k, _, _ in flat.groups]
        assert keys == ["event_fc", "conv_blocks.3", "conv_blocks.2", "conv_blocks.1", "conv_blocks.0"]
        # default: two buckets -- head (>= 85 % of the elements, late layers) and tail
        assert [list(k) for k, _, _ in flat.buckets] == [["event_fc", "conv_blocks.3", "conv_blocks.2"],
                                                          ["conv_blocks.1", "conv_blocks.0"]]
        assert [list(k) for k, _, _ in tr.FlatParams._merge_buckets(flat.groups, 1)] == [keys]
        assert len(tr.FlatParams._merge_buckets(flat.groups, 0)) == 5
        for nb in (0, 1, 2, 3):
            covered = sorted((s, e) for _, s, e in tr.FlatParams._merge_buckets(flat.groups, nb))
            assert covered[0][0] == 0 and covered[-1][1] == flat.numel
            assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
        red = tr.GradAllReducer(flat.g, flat.buckets)
        assert red.enabled and red.world == world
        for n in flat.names:                             # rank-dependent gradients
            flat.G[n].fill_(float(rank + 1))
        for i, k in enumerate(keys):                      # backward completion order
            red.bucket_ready(k)
            assert red.issued == ([] if i < 2 else [0] if i < 4 else [0, 1])      # a bucket leaves with its LAST group
        scale = red.finish()
        avg = flat.g * scale
        ok = bool(torch.allclose(avg[: 4 * 1 * 9], torch.full((36,), (1 + 2) / 2.0)))
        for n in flat.names:
            ok &= bool(torch.allclose(flat.G[n] * scale, torch.full_like(flat.G[n], 1.5)))
        with pytest.raises(KeyError):
            red.bucket_ready("nope")
        # replicas: identical host RNG state after seed_all_ranks (rank 0's draw wins), rank-sharded disjoint batches
        import random
        random.seed(1000 + rank); np.random.seed(1000 + rank); torch.manual_seed(1000 + rank)
        seed = tr.seed_all_ranks()
        draws = (random.random(), float(np.random.rand()), float(torch.rand(1)))
        ds = importlib.import_module("soundeventdetection-pytorch_amd.dataset.spectogram.spectograms_dataset")
        names = [f"rec_{i:02d}" for i in range(20)]
        split = ds.split_train_val(list(names), 0.2)       # shuffled percentage split (the reference's default)
        gathered = [None, None]
        dist.all_gather_object(gathered, (seed, draws, split))
        ok &= (gathered[0] == gathered[1]) or print('RNG/split mismatch', gathered) is not None
        syn = importlib.import_module("soundeventdetection-pytorch_amd.dataset.synthetic")
        loader = tr.ShardedBatchLoader(syn.SyntheticSedDataset(n_train_crops=22, crop=16, n_val=1, val_frames=16), 4, rank, world)
        mine = [i for idx in loader.indices() for i in idx]
        nsteps = len(list(loader.indices()))
        both = [None, None]
        dist.all_gather_object(both, (mine, nsteps))
        steps = (22 + 7) // 8
        a, b = both[0][0], both[1][0]
        # equal per-rank batches (1/world average, SyncBN count * world); nothing dropped: the ragged tail of the last global
        # batch wraps to the head of the table, the same rule on both ranks
        ok &= (len(a) == len(b) == steps * 4 and both[0][1] == both[1][1] == len(loader) == steps
               and set(a) | set(b) == set(range(22)) and not (set(a[:8]) & set(b[:8]))
               and a[8:] == [16, 17, 18, 19] and b[8:] == [20, 21, 0, 1]) or print('shards', both) is not None
        tr.reseed_rank(seed, rank)
        after = [None, None]
        dist.all_gather_object(after, (random.random(), float(np.random.rand()), float(torch.rand(1))))
        ok &= (after[0] != after[1]) or print('rank streams identical after reseed_rank', after) is not None
        xb, yb = next(iter(loader))
        ok &= (tuple(xb.shape) == (4, 1, 16, 64) and tuple(yb.shape) == (4, 16, 1)) or print('shapes', xb.shape, yb.shape) is not None
        q.put((rank, ok, scale))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in res) == [0, 1]
    assert all(ok for _, ok, _ in res) and all(s == 0.5 for _, _, s in res)


def test_single_process_reducer_is_noop():
    tr = importlib.import_module("soundeventdetection-pytorch_amd.train")
    g = torch.ones(8)
    red = tr.GradAllReducer(g, [("a", 0, 8)])
    assert not red.enabled
    red.bucket_ready("a")
    assert red.finish() == 1.0 and torch.equal(g, torch.ones(8))


def test_bench_launcher_with_eight_ranks_rendezvous_only():
    """The driver's 8-GPU command shape, `python bench.py --gpus 8 ...`, with EIGHT ranks: the parent starts them itself
    (torch.distributed.run as a child), they rendezvous on 127.0.0.1, run the barrier / timed region / barrier / MAX-over-ranks of a bench
    run around an empty step and rank 0's single line comes back on the parent's stdout.  Over gloo and without the GPU
    (`--rendezvous-only`): a GPU box admits six processes on its card, so the 8-rank launch itself is rehearsed here and the train
    step under several ranks in tests/test_gpu_ddp.py (2 and 4 ranks).  SURVEY 8(e); replaces the reference's single-device launch
    /root/reference/main.py:121-136."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--batch", "1", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--rendezvous-only"]
    p = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["steps"] == 3 and r["config"]["global_batch"] == 8 and r["config"]["parallelism"] == "dp8"
    assert r["config"]["max_over_ranks_ok"] is True and "rehearsal" in r
    # the slowest rank sleeps 8 ms per step: the reported time is the MAX over ranks, not rank 0's
    assert r["ms_per_step"] >= 8.0
    err = p.stderr.decode(errors="replace")
    assert all(f"rank {k}/8: rendezvous ok" in err for k in range(8))
