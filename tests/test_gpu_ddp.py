"""Data-parallel train step on real kernels: two fresh child processes (tests/ddp_gpu_worker.py), both on cuda:0,
process group gloo (several ranks cannot share one GPU under RCCL).  Reference semantics: ONE optimizer step over the
global batch (/root/reference/train.py:95-103); with SyncBN also global-batch BatchNorm statistics
(models/spectogram_models.py:142-143 executed by one process).  fp32 mode, so the comparisons are tight."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "ddp_gpu_worker.py")


def _run(world, mode, out, port, extra_env=None):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.pop("SED_DDP_BUCKETS", None)
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), mode, out], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for p, l in zip(procs, logs):
        assert p.returncode == 0, l[-3000:]
    return logs


@pytest.fixture(scope="module")
def port():
    return 29600 + os.getpid() % 300


def test_identical_shards_reproduce_single_process_bitwise(tmp_path, port):
    assert torch.cuda.is_available()
    solo, duo = str(tmp_path / "solo.pt"), str(tmp_path / "duo.pt")
    _run(1, "same", solo, port)
    _run(2, "same", duo, port + 1)
    a, b = torch.load(solo), torch.load(duo + ".r0")
    assert b["buckets"] == [["event_fc", "conv_blocks.3", "conv_blocks.2"], ["conv_blocks.1", "conv_blocks.0"]]
    assert a["losses"] == b["losses"]
    assert torch.equal(a["p"], b["p"])           # (g + g) / 2 == g exactly: three Adam-amsgrad steps stay bit-identical
    for k in a["sd"]:
        assert torch.equal(a["sd"][k], b["sd"][k]), k


def test_different_shards_average_the_rank_gradients(tmp_path, port):
    outs = {}
    for r in (0, 1):
        outs[r] = str(tmp_path / f"solo{r}.pt")
        _run(1, f"solo:{r}", outs[r], port + 2)
    duo = str(tmp_path / "duo.pt")
    _run(2, "shard", duo, port + 3)
    g0, g1 = torch.load(outs[0])["g"], torch.load(outs[1])["g"]
    r0, r1 = torch.load(duo + ".r0"), torch.load(duo + ".r1")
    assert torch.equal(r0["g"], r1["g"])          # every rank ends with the same averaged gradient
    np.testing.assert_allclose(r0["g"].numpy(), ((g0.double() + g1.double()) / 2).numpy(), rtol=1e-6, atol=1e-9)
    assert r0["issued"] == [0, 1]                 # head bucket (issued while blocks 1/0 still ran backward), then the tail
    # rank 1's replica started from a different seed: the constructor broadcast made it rank 0's
    for k in r0["sd"]:
        if not k.endswith("running_mean") and not k.endswith("running_var"):
            assert torch.equal(r0["sd"][k], r1["sd"][k]), k


def test_syncbn_two_half_batches_equal_one_full_batch(tmp_path, port):
    solo, duo = str(tmp_path / "full.pt"), str(tmp_path / "sync.pt")
    _run(1, "shard", solo, port + 4)              # world 1: the whole batch in one process
    _run(2, "sync", duo, port + 5)
    full, r0, r1 = torch.load(solo), torch.load(duo + ".r0"), torch.load(duo + ".r1")
    B = full["logits"].shape[0]
    lg = torch.cat([r0["logits"], r1["logits"]], 0)
    assert lg.shape[0] == B
    np.testing.assert_allclose(lg.numpy(), full["logits"].numpy(), atol=1e-3, rtol=0)       # north_star's fp32 gate
    np.testing.assert_allclose((r0["loss"] + r1["loss"]) / 2, full["loss"], rtol=1e-5)
    gs, gf = r0["g"].double(), full["g"].double()
    assert torch.equal(r0["g"], r1["g"])
    rel = float((gs - gf).norm() / gf.norm())
    assert rel < 1e-3, rel
    for k, v in full["sd"].items():               # running statistics of the GLOBAL batch on every rank
        if k.endswith("running_mean") or k.endswith("running_var"):
            np.testing.assert_allclose(r0["sd"][k].numpy(), v.numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
            assert torch.equal(r0["sd"][k], r1["sd"][k]), k


def test_syncbn_bf16_identical_batches_reproduce_the_single_process_gradient(tmp_path, port, monkeypatch):
    """The throughput-mode arithmetic under SyncBN: bf16, C1 mode, pool-backward statistics from the data-gradient epilogue
    (their per-rank partial rows are what the ranks all-reduce).  Both ranks hold the SAME full batch, so every global sum is
    exactly twice the local one and the gradient must be that of a single process that reduces its statistics the same way
    (row sums in fp32, then the finalize) -- a sharp check of the plumbing.  Against the ordinary single-process path (fp64
    sum of all partial rows) this tiny bf16 network amplifies the 1e-7 difference of the first BatchNorm's scale about 5x
    per layer, to 0.6 % of the logits and 7 % of the gradient norm (tools/sync_probe.py; the same with half batches, without
    the fused statistics and without C1 mode): not a property of the data-parallel path."""
    monkeypatch.setenv("SED_TEST_PRECISION", "bf16")
    solo, duo = str(tmp_path / "full.pt"), str(tmp_path / "sync.pt")
    _run(1, "samesync", solo, port + 6)
    _run(2, "samesync", duo, port + 7)
    full, r0, r1 = torch.load(solo), torch.load(duo + ".r0"), torch.load(duo + ".r1")
    np.testing.assert_allclose(r0["logits"].numpy(), full["logits"].numpy(), atol=1e-5, rtol=0)
    gs, gf = r0["g"].double(), full["g"].double()
    assert torch.equal(r0["g"], r1["g"])
    rel = float((gs - gf).norm() / gf.norm())
    assert rel < 1e-4, rel


def test_rccl_executes_the_collectives_on_a_world_size_1_group(tmp_path, port):
    """torch.distributed backend "nccl" (= RCCL on ROCm) cannot put two ranks on the box's one GPU, but a world-size-1 group can
    run the REAL collectives: with SED_DDP_FORCE=1 the trainer treats it like any other group -- the flat-parameter and buffer
    broadcast, the two bucketed async all-reduces issued from inside the backward, their .wait() before Adam, and (samesync)
    the per-layer SyncBN all-reduces ordered on the compute stream.  A sum over one rank is the identity, so three train steps
    must reproduce the run without a process group bit for bit; and the SyncBN step must equal the single-process twin that
    reduces its statistics the same way."""
    rccl = {"SED_TEST_BACKEND": "nccl", "SED_DDP_FORCE": "1"}
    solo, one = str(tmp_path / "solo.pt"), str(tmp_path / "rccl.pt")
    _run(1, "same", solo, port + 8)
    logs = _run(1, "same", one, port + 9, rccl)
    assert "process group backend: nccl world 1 reducer enabled: True" in logs[0], logs[0][-2000:]
    a, b = torch.load(solo), torch.load(one)
    assert a["losses"] == b["losses"]
    assert torch.equal(a["p"], b["p"])
    for k in a["sd"]:
        assert torch.equal(a["sd"][k], b["sd"][k]), k
    twin, sync = str(tmp_path / "twin.pt"), str(tmp_path / "rccl_sync.pt")
    _run(1, "samesync", twin, port + 10, {"SED_TEST_PRECISION": "bf16"})
    logs = _run(1, "samesync", sync, port + 11, dict(rccl, SED_TEST_PRECISION="bf16"))
    assert "bn_sync: BnSync" in logs[0], logs[0][-2000:]
    t, r = torch.load(twin), torch.load(sync)
    assert r["issued"] == [0, 1]
    assert torch.equal(t["logits"], r["logits"]) and torch.equal(t["g"], r["g"])


@pytest.mark.parametrize("sync_bn", [0, 1])
def test_bench_launches_its_own_ranks(sync_bn):
    """`python bench.py --gpus 2` as a PLAIN command (the driver's command shape; replaces the reference's single-device launch
    /root/reference/main.py:121-136): the parent starts two ranks itself (torch.distributed.run as a child, both on cuda:0 over
    gloo), relays exactly one JSON line and returns rc 0.  Exercises the rank > 0 code of bench.py: stdout redirection, the
    barriers around the timed region, the MAX-reduce of the elapsed time, the collectives inside the step."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SED_DDP_FORCE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--batch", "2", "--seconds", "4",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--sync-bn", str(sync_bn)]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1
    assert r["config"]["global_batch"] == 4 and r["config"]["parallelism"] == "dp2"
    assert r["config"]["collectives_executed"] is True
    assert r["config"]["sync_bn"] is bool(sync_bn)
    assert r["value"] > 0 and np.isfinite(r["loss"])
    assert abs(r["value"] - 4 * 2 / (r["ms_per_step"] * 2 / 1e3)) < 1e-6 * r["value"]
    assert "cpu_baseline" not in r                      # N > 1: rank 0 reports no CPU leg
    # round 6: the line says where a data-parallel step's communication went -- every bucket's all-reduce in isolation, the time the
    # compute stream was blocked on the collectives after the backward (exposed communication), the SyncBN collectives per step
    comm = r["config"]["comm"]
    assert [b["groups"] for b in comm["buckets"]] == r["config"]["grad_buckets"] and len(comm["buckets"]) == 2
    assert sum(b["floats"] for b in comm["buckets"]) == comm["flat_single_allreduce"]["floats"] >= 582433
    assert all(b["us_per_allreduce_isolated"] > 0 for b in comm["buckets"]) and comm["flat_single_allreduce"]["us_per_allreduce_isolated"] > 0
    assert comm["exposed_ms_per_step"] is not None and 0 <= comm["exposed_share_of_step"] < 1
    assert (comm["syncbn_collectives_per_step"] > 0) is bool(sync_bn)


def test_bench_allreduce_only_on_the_world_1_rccl_group():
    """`bench.py --allreduce-only` (round 6): the two-bucket layout of the 2.33 MB flat gradient buffer and one flat all-reduce, timed in
    isolation over RCCL -- on the one GPU of this box as a world-size-1 group (SED_DDP_FORCE=1 under torch.distributed.run
    --nproc-per-node 1: librccl's kernels really run), the same command an 8-GPU node would take with --nproc-per-node 8.  SURVEY 8(e)."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["SED_DDP_FORCE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr=127.0.0.1", "--master-port=29571",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--allreduce-only", "--steps", "50"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["rccl_group"] is True and r["n_gpus"] == 1 and r["unit"] == "us" and r["higher_is_better"] is False
    assert len(r["buckets"]) == 2 and r["flat"]["bytes"] == sum(b["bytes"] for b in r["buckets"]) >= 582433 * 4
    assert 0 < r["value"] < 5000 and all(0 < b["us_per_allreduce_isolated"] < 5000 for b in r["buckets"])


def test_bench_launches_four_ranks_on_one_gpu():
    """`python bench.py --gpus 4` as a plain command with FOUR ranks sharing cuda:0 over gloo (a GPU box admits six processes on its card and
    this pytest process is one of them; the 8-rank launch itself is rehearsed without the GPU: tests/test_ddp_gloo.py) -- one clip of 2 s per
    rank: the train step, its gradient all-reduces and the timed region's barriers / MAX-reduce under more than two ranks; one JSON line with
    `n_gpus: 4`, `global_batch: 4`.  SURVEY 8(e)."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SED_DDP_FORCE"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--batch", "1", "--seconds", "2",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 4 and r["config"]["global_batch"] == 4 and r["config"]["parallelism"] == "dp4"
    assert r["config"]["collectives_executed"] is True
    assert r["value"] > 0 and np.isfinite(r["loss"])
    assert abs(r["value"] - 4 * 1 / (r["ms_per_step"] / 1e3)) < 1e-6 * r["value"]
