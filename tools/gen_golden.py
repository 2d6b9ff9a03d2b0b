#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the real reference (runs only in the build container,
where /root/reference exists; the fixtures are data and are committed, the reference never is).

    python tools/gen_golden.py            # rewrites tests/golden/

Fixture list follows SURVEY.md 8(c): G1 ConvBlock fwd+bwd, G2 Cnn_AvgPooling train steps + Adam,
G3 eval-mode forward + decisions/onsets, G4 WeightedBCE, G5 metrics, G6 interpolate, G7 M5,
G8 loss trace of the reference train() loop.  All tensors float32 unless noted.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

import matplotlib  # noqa: E402
matplotlib.use("Agg")

from models.spectogram_models import Cnn_AvgPooling, ConvBlock, interpolate  # noqa: E402
from utils.common import WeightedBCE  # noqa: E402
from utils.metric_utils import calculate_metrics, compute_recall_precision, f_score  # noqa: E402
import train as ref_train  # noqa: E402

MAIN_CFG = [(32, 2), (64, 2), (128, 2), (128, 1)]
TINY_CFG = [(4, 2), (8, 2), (8, 2), (8, 1)]

torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def state_np(m, prefix="sd."):
    return {prefix + k: npy(v) for k, v in m.state_dict().items()}


def make_targets(gen, B, T, K, p=0.04, run=10):
    """Bernoulli(p) events in runs of >= `run` frames."""
    y = np.zeros((B, T, K), dtype=np.float32)
    n_runs = max(1, int(round(p * T / run)))
    for b in range(B):
        for k in range(K):
            for _ in range(n_runs):
                s = int(gen.integers(0, max(1, T - run)))
                y[b, s:s + run + int(gen.integers(0, run)), k] = 1.0
    return y


def onset_idx(dec):
    d = np.diff(np.concatenate([[0], dec.astype(np.int8)]))
    return np.flatnonzero(d == 1)


def g1_convblock():
    """G1: ConvBlock train-mode fwd + autograd bwd, tiny widths (full tensors) for pool 2 and 1,
    odd H to exercise the floor of avg_pool2d."""
    out = {}
    for tag, (cin, cout, pool, B, H, W) in {"a": (1, 4, 2, 2, 13, 64), "b": (4, 8, 2, 2, 7, 16),
                                              "c": (8, 8, 1, 3, 5, 8)}.items():
        torch.manual_seed(100 + ord(tag))
        blk = ConvBlock(cin, cout, pool)
        with torch.no_grad():   # non-trivial BN affine so dgamma/dbeta matter
            for bn in (blk.bn1, blk.bn2):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.uniform_(-0.3, 0.3)
        blk.train()
        x = torch.randn(B, cin, H, W, requires_grad=True)
        sd0 = {f"{tag}.sd0.{k}": npy(v).copy() for k, v in blk.state_dict().items()}
        y = blk(x)
        dy = torch.randn_like(y)
        y.backward(dy)
        out.update(sd0)
        out.update({f"{tag}.sd1.{k}": npy(v) for k, v in blk.state_dict().items()})
        out.update({f"{tag}.x": npy(x), f"{tag}.y": npy(y), f"{tag}.dy": npy(dy), f"{tag}.dx": npy(x.grad),
                    f"{tag}.pool": np.int64(pool)})
        for n, p in blk.named_parameters():
            out[f"{tag}.grad.{n}"] = npy(p.grad)
    np.savez_compressed(os.path.join(OUT, "g1_convblock.npz"), **out)


def g2_train_steps():
    """G2: Cnn_AvgPooling (tiny config: full tensors; main config: slices + norms), train-mode
    logits, loss, grads, and parameters after Adam-amsgrad steps across an LR-decay boundary."""
    out = {}
    gen = np.random.default_rng(7)
    for tag, cfg, K, B, T in (("tiny13", TINY_CFG, 1, 4, 13), ("tiny30k3", TINY_CFG, 3, 2, 30),
                              ("main13", MAIN_CFG, 1, 4, 13), ("main30", MAIN_CFG, 1, 4, 30)):
        torch.manual_seed(0)
        m = Cnn_AvgPooling(K, model_config=cfg)
        m.train()
        crit = WeightedBCE(recall_factor=5, multi_frame=True)
        x = torch.randn(B, 1, T, 64)
        y = torch.from_numpy(make_targets(gen, B, T, K, p=0.2, run=3))
        full = tag.startswith("tiny")
        if full:
            out.update({f"{tag}.sd0.{k}": npy(v).copy() for k, v in m.state_dict().items()})
        out[f"{tag}.x"], out[f"{tag}.y"] = npy(x), npy(y)
        # lr large enough that 3 steps visibly move the weights; decay boundary forced at step 2
        lr = 1e-3
        opt = torch.optim.Adam(m.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.,
                               amsgrad=True)
        for step in range(1, 4):
            logits = m(x)
            loss = crit(logits, y)
            opt.zero_grad()
            loss.backward()
            if step == 1:
                out[f"{tag}.logits"] = npy(logits)
                out[f"{tag}.loss"] = npy(loss)
                for n, p in m.named_parameters():
                    g = npy(p.grad)
                    out[f"{tag}.gnorm.{n}"] = np.float64(np.linalg.norm(g.astype(np.float64)))
                    if full:
                        out[f"{tag}.grad.{n}"] = g
                    else:
                        out[f"{tag}.gslice.{n}"] = g.reshape(-1)[:64].copy()
            opt.step()
            out[f"{tag}.loss_step{step}"] = npy(loss)
            if step == 2:   # emulate train.py:108-110 with lr_decay_freq = 2
                for pg in opt.param_groups:
                    pg["lr"] *= 0.997
            if step in (1, 3):
                for n, p in m.named_parameters():
                    v = npy(p)
                    if full:
                        out[f"{tag}.p_step{step}.{n}"] = v.copy()
                    else:
                        out[f"{tag}.pslice_step{step}.{n}"] = v.reshape(-1)[:64].copy()
                        out[f"{tag}.pnorm_step{step}.{n}"] = np.float64(np.linalg.norm(v.astype(np.float64)))
        if full:
            out.update({f"{tag}.sd3.{k}": npy(v) for k, v in m.state_dict().items()
                        if "running" in k or "tracked" in k})
        else:
            for k, v in m.state_dict().items():
                if "running" in k:
                    out[f"{tag}.sd3slice.{k}"] = npy(v)[:16].copy()
    np.savez_compressed(os.path.join(OUT, "g2_train_steps.npz"), **out)


def g3_eval_forward():
    """G3: eval-mode forward with non-trivial running stats, main config, B=1, T=182 and 6001:
    logit slices, checksum, threshold decisions and onset indices."""
    out = {}
    torch.manual_seed(0)
    m = Cnn_AvgPooling(1, model_config=MAIN_CFG)
    # make running statistics realistic: a few train-mode passes, then freeze
    m.train()
    with torch.no_grad():
        for _ in range(3):
            m(torch.randn(4, 1, 64, 64))
        m.event_fc.bias.fill_(0.02)
    m.eval()
    out["seed_note"] = np.array("torch.manual_seed(0); 3 train passes randn(4,1,64,64); fc.bias=.02")
    out.update({f"sd.{k}": npy(v) for k, v in m.state_dict().items() if "num_batches" not in k})
    for T in (182, 6001):
        g = torch.Generator().manual_seed(1000 + T)
        x = torch.randn(1, 1, T, 64, generator=g)
        # burst structure so that decisions flip along time
        env = torch.zeros(T)
        for s in range(20, T - 40, max(40, T // 12)):
            env[s:s + 24] = 2.5
        x = x + env[None, None, :, None]
        with torch.no_grad():
            lg = m(x)[0, :, 0]
        lg = npy(lg)
        out[f"T{T}.seed"] = np.int64(1000 + T)
        out[f"T{T}.logits"] = lg
        out[f"T{T}.decisions"] = (lg > 0)
        out[f"T{T}.onsets"] = onset_idx(lg > 0)
        if T == 182:
            out[f"T{T}.x"] = npy(x)
    np.savez_compressed(os.path.join(OUT, "g3_eval_forward.npz"), **out)


def g4_bce():
    out = {}
    torch.manual_seed(4)
    for tag, (B, To, Tt, K, w) in {"trunc_out_longer": (3, 24, 20, 1, 5.0), "trunc_tgt_longer": (3, 24, 30, 1, 5.0),
                                    "k3": (2, 16, 16, 3, 2.0), "w1": (2, 8, 8, 1, 1.0)}.items():
        o = (torch.randn(B, To, K) * 3).requires_grad_()
        t = (torch.rand(B, Tt, K) > 0.7).float()
        loss = WeightedBCE(w, True)(o, t)
        loss.backward()
        out.update({f"{tag}.o": npy(o), f"{tag}.t": npy(t), f"{tag}.w": np.float64(w), f"{tag}.loss": npy(loss),
                    f"{tag}.do": npy(o.grad)})
    o = torch.randn(7, 1, requires_grad=True)
    t = (torch.rand(7) > 0.5).float()
    loss = WeightedBCE(5, False)(o, t)
    loss.backward()
    out.update({"single.o": npy(o), "single.t": npy(t), "single.w": np.float64(5), "single.loss": npy(loss),
                "single.do": npy(o.grad)})
    np.savez_compressed(os.path.join(OUT, "g4_bce.npz"), **out)


def g5_metrics():
    out = {}
    gen = np.random.default_rng(5)
    cases = {
        "rand": (gen.random((200, 1)).astype(np.float32), (gen.random((200, 1)) > 0.8).astype(np.float32)),
        "no_gt": (gen.random((50, 1)).astype(np.float32), np.zeros((50, 1), np.float32)),
        "all_gt": (gen.random((50, 1)).astype(np.float32), np.ones((50, 1), np.float32)),
        "len_mismatch": (gen.random((64, 1)).astype(np.float32), (gen.random((61, 1)) > 0.5).astype(np.float32)),
        "k3": (gen.random((80, 3)).astype(np.float32), (gen.random((80, 3)) > 0.6).astype(np.float32)),
        "edges": (np.array([[0.0], [1.0], [0.05], [0.5], [0.95], [1.0], [0.0]], np.float32),
                  np.array([[0], [1], [1], [0], [1], [1], [0]], np.float32)),
    }
    for tag, (o, t) in cases.items():
        r, p, ap = calculate_metrics(o, t)
        out.update({f"{tag}.o": o, f"{tag}.t": t, f"{tag}.recalls": r, f"{tag}.precisions": p, f"{tag}.AP": np.float64(ap)})
        # ProgressPlotter.report_validation_metrics call convention (common.py:53-54)
        out[f"{tag}.f1"] = f_score(p, r, precision_importance_factor=1)
        out[f"{tag}.f5"] = f_score(p, r, precision_importance_factor=5)
    O = (gen.random((30, 2)) > 0.5).astype(np.int64)
    T = (gen.random((30, 2)) > 0.5).astype(np.float32)
    rc, pr = compute_recall_precision(O, T)
    out.update({"crp.O": O, "crp.T": T, "crp.recall": np.float64(rc), "crp.prec": np.float64(pr)})
    np.savez_compressed(os.path.join(OUT, "g5_metrics.npz"), **out)


def g6_interpolate():
    torch.manual_seed(6)
    x = torch.randn(2, 5, 3)
    np.savez_compressed(os.path.join(OUT, "g6_interpolate.npz"), x=npy(x), r8=npy(interpolate(x, 8)),
                        r2=npy(interpolate(x, 2)), r1=npy(interpolate(x, 1)))


class _ListDataset(torch.utils.data.Dataset):
    def __init__(self, xs, ys):
        self.xs, self.ys = xs, ys

    def __len__(self):
        return len(self.xs)

    def __getitem__(self, i):
        return self.xs[i], self.ys[i]


def g8_train_trace():
    """G8: loss trace of the reference train() (train.py:77-131) for 6 steps over an in-memory
    list dataset, log_freq > num_steps so no eval/plots/checkpoints run.  lr_decay_freq is the
    reference's 200, so no decay inside the trace; the decay boundary is covered by G2."""
    import tempfile
    gen = np.random.default_rng(8)
    K, T, N, bs = 1, 30, 8, 4
    xs = [torch.from_numpy(gen.standard_normal((1, T, 64)).astype(np.float32)) for _ in range(N)]
    ys = [torch.from_numpy(make_targets(gen, 1, T, K, p=0.2, run=3)[0].astype(np.float64)) for _ in range(N)]
    torch.manual_seed(0)
    m = Cnn_AvgPooling(K, model_config=TINY_CFG)
    sd0 = {f"sd0.{k}": npy(v).copy() for k, v in m.state_dict().items()}
    dl = torch.utils.data.DataLoader(_ListDataset(xs, ys), batch_size=bs)
    losses = []
    orig = ref_train.ProgressPlotter.report_train_loss

    def rec(self, loss):
        losses.append(loss)
        return orig(self, loss)
    ref_train.ProgressPlotter.report_train_loss = rec
    with tempfile.TemporaryDirectory() as d:
        ref_train.train(m, dl, WeightedBCE(5, True), num_steps=6, lr=1e-3, log_freq=1000, outputs_dir=d,
                        device=torch.device("cpu"))
    ref_train.ProgressPlotter.report_train_loss = orig
    out = dict(sd0)
    out.update({f"sd6.{k}": npy(v) for k, v in m.state_dict().items()})
    out["x"] = np.stack([npy(x) for x in xs])
    out["y"] = np.stack([npy(y) for y in ys])
    out["losses"] = np.array(losses, dtype=np.float64)
    out["batch_size"] = np.int64(bs)
    out["lr"] = np.float64(1e-3)
    np.savez_compressed(os.path.join(OUT, "g8_train_trace.npz"), **out)


def g7_m5():
    """G7: raw-waveform M5 (models/waveform_models.py:9-71), seed 0, B=8: train-mode logits, loss
    (WeightedBCE(5, multi_frame=False)), every parameter gradient, BN buffers after the step, parameters
    after 1 and 3 Adam-amsgrad steps (lr 1e-3) at L=2048; eval-mode logits at the reference frame size
    L=31680 from a seeded input (regenerated by the tests: torch.manual_seed(7); randn(8, 1, 31680) * 0.1)."""
    from models.waveform_models import M5
    gen = np.random.default_rng(7)
    B, L = 8, 2048
    x = torch.from_numpy((gen.standard_normal((B, 1, L)) * 0.1).astype(np.float32))
    y = torch.from_numpy((gen.random(B) > 0.6).astype(np.float32))
    torch.manual_seed(0)
    m = M5(1)
    out = {f"sd0.{k}": npy(v).copy() for k, v in m.state_dict().items()}
    out["x"], out["y"] = npy(x), npy(y)

    def sample(prefix, k, a):
        """small tensors in full; large ones as norm + head + strided sample (keeps the fixture small)"""
        a = np.asarray(a)
        if a.size <= 4096:
            out[f"{prefix}.{k}"] = a.copy()
        else:
            f = a.reshape(-1)
            out[f"{prefix}_norm.{k}"] = np.float64(np.sqrt((f.astype(np.float64) ** 2).sum()))
            out[f"{prefix}_head.{k}"] = f[:512].copy()
            out[f"{prefix}_stride.{k}"] = f[::97].copy()
    crit = WeightedBCE(5, False)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0., amsgrad=True)
    m.train()
    for step in range(1, 4):
        opt.zero_grad()
        logits = m(x)
        loss = crit(logits, y)
        loss.backward()
        if step == 1:
            out["logits"], out["loss"] = npy(logits), np.float64(loss.item())
            for k, p_ in m.named_parameters():
                sample("grad", k, npy(p_.grad))
        opt.step()
        if step in (1, 3):
            for k, v in m.state_dict().items():
                sample(f"sd{step}", k, npy(v))
    out["loss3"] = np.float64(loss.item())
    # eval-mode forward at the reference frame size with the step-3 weights / running statistics
    m.eval()
    torch.manual_seed(7)
    xe = torch.randn(8, 1, 31680) * 0.1
    with torch.no_grad():
        out["eval_logits_31680"] = npy(m(xe))
    np.savez_compressed(os.path.join(OUT, "g7_m5.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    for fn in (g1_convblock, g2_train_steps, g3_eval_forward, g4_bce, g5_metrics, g6_interpolate, g7_m5, g8_train_trace):
        if only and fn.__name__ not in only:
            continue
        fn()
        print("wrote", fn.__name__)
    for f in sorted(os.listdir(OUT)):
        print(f"{f:32s} {os.path.getsize(os.path.join(OUT, f)) / 1024:8.1f} KB")
