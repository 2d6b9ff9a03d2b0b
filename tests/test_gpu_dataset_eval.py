"""Data-side and eval-side kernels on the MI355X vs oracle/dataset_oracle.py, through the C ABI:
crop gather + z-score ('logMel'), gather + mix + noise + complex z-score + log-mel ('Complex'),
the 21-threshold counting of calculate_metrics (bit-exact integers), SpectogramDataset /
DeviceBatchLoader end to end, train() + eval() on it, and the inference CLI."""
import ctypes
import importlib
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import dataset_oracle as DO
from oracle import frontend_oracle as FO

from conftest import load_golden

pytestmark = pytest.mark.gpu
PKG = "soundeventdetection-pytorch_amd"


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available()
    L = importlib.import_module(PKG + "._lib")
    sc = importlib.import_module(PKG + ".dataset.spectogram.spectogram_configs")
    pp = importlib.import_module(PKG + ".dataset.spectogram.preprocess")
    return L, sc, pp


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_metric_counts_bit_exact(env):
    L, _, _ = env
    mu = importlib.import_module(PKG + ".utils.metric_utils")
    g = load_golden("g5_metrics.npz")
    # probabilities given (no sigmoid): counts, recalls, precisions and AP equal the reference's exactly
    for tag in ("rand", "no_gt", "all_gt", "len_mismatch", "k3", "edges"):
        o, t = g[f"{tag}.o"].astype(np.float32), g[f"{tag}.t"].astype(np.float32)
        tp, pos, gt = mu.metric_counts_device(dev(o), dev(t))
        etp, epos, egt = DO.threshold_counts(o, t)
        assert np.array_equal(tp.cpu().numpy(), etp) and np.array_equal(pos.cpu().numpy(), epos), tag
        assert float(gt.cpu()) == egt
        r, p, ap = mu.calculate_metrics_device(dev(o), dev(t))
        assert np.array_equal(r, g[f"{tag}.recalls"]) and np.array_equal(p, g[f"{tag}.precisions"]), tag
        assert ap == float(g[f"{tag}.AP"])
    # raw logits: the device sigmoid feeds the same counting; against the oracle on the device's own
    # probabilities it is exact, against torch's CPU sigmoid at most a handful of threshold ties differ
    rng = np.random.default_rng(5)
    x = (3 * rng.standard_normal((200003, 1))).astype(np.float32)          # > 256 blocks * 256 threads
    t = (rng.random((200000, 1)) < 0.1).astype(np.float32)
    t[7] = 0.5                                                              # exercises the (2T - 0) == 1 branch
    tp, pos, gt, probs = mu.metric_counts_device(dev(x), dev(t), raw_logits=True, return_probs=True)
    pr = probs.cpu().numpy()
    assert pr.shape == (200000, 1)
    etp, epos, egt = DO.threshold_counts(pr, t)
    assert np.array_equal(tp.cpu().numpy(), etp) and np.array_equal(pos.cpu().numpy(), epos)
    assert float(gt.cpu()) == egt
    ctp, cpos, _ = DO.threshold_counts(torch.sigmoid(torch.from_numpy(x)).numpy(), t)
    assert np.abs(tp.cpu().numpy() - ctp).max() <= 3 and np.abs(pos.cpu().numpy() - cpos).max() <= 3
    np.testing.assert_allclose(pr, torch.sigmoid(torch.from_numpy(x[:200000])).numpy(), atol=2e-7)
    # descending thresholds are refused before any launch
    ths = (ctypes.c_double * 2)(0.5, 0.1)
    ws = torch.empty(L.lib().sed_metric_counts_ws_bytes(2) // 8 + 1, dtype=torch.float64, device="cuda")
    c = torch.zeros(2, 2, dtype=torch.int64, device="cuda")
    gs = torch.zeros(1, dtype=torch.float64, device="cuda")
    rc = L.lib().sed_metric_counts(L.ptr(dev(x)), L.ptr(dev(t)), None, ctypes.cast(ths, ctypes.c_void_p), 2, 1,
                                   L.ptr(c), L.ptr(gs), L.ptr(ws), 10, 10, 1, None)
    assert rc != 0 and b"ascending" in L.lib().sed_last_error()


def _complex_bank(rng, frames, bins):
    return (rng.standard_normal((frames, bins)) + 1j * rng.standard_normal((frames, bins))).astype(np.complex64)


@pytest.mark.parametrize("which", ["BENCH", "REF_NATIVE"])
def test_complex_augment_logmel_matches_oracle(env, which):
    L, sc, pp = env
    cfg = getattr(sc, which)
    ocfg = FO.FrontEndConfig(cfg.working_sample_rate, cfg.frame_size, cfg.hop_size, cfg.NFFT)
    mel = FO.mel_filter_bank_matrix(ocfg)
    fe = pp.LogMelFrontEnd(cfg, "cuda")
    rng = np.random.default_rng(11)
    bins, crop, B, frames = cfg.bins, 9, 5, 64
    bank = _complex_bank(rng, frames, bins) * np.float32(30.0)
    events = (rng.random((frames, 1)) < 0.2).astype(np.float64)
    cmean = (0.1 * (rng.standard_normal(bins) + 1j * rng.standard_normal(bins))).astype(np.complex64)
    cstd = (1.0 + rng.random(bins)).astype(np.float32)
    starts = np.array([[0, 0, 0, 0], [5, 40, 0, 0], [55, 1, 17, 0], [3, 9, 27, 50], [20, 0, 0, 0]], dtype=np.int32)
    nmix = np.array([1, 2, 3, 4, 1], dtype=np.int32)
    nstd = np.array([0.0, 0.004, 0.0, 0.0065, 0.006], dtype=np.float32)
    z = rng.standard_normal((B, crop, bins)).astype(np.float32)
    out = torch.empty((B, crop, 64), dtype=torch.float32, device="cuda")

    def run(noise, seed, mean, std):
        L.check(L.lib().sed_complex_augment_logmel(
            L.ptr(d_bank), frames, starts.ctypes.data, nmix.ctypes.data, L.ptr(d_starts), L.ptr(d_nmix), L.ptr(d_nstd),
            L.ptr(noise), ctypes.c_ulonglong(seed), L.ptr(mean), L.ptr(std), L.ptr(fe.melT), L.ptr(fe.mel_lo),
            L.ptr(fe.mel_hi), L.ptr(out), B, crop, bins, 64, None), "complex_augment_logmel")
        torch.cuda.synchronize()
        return out.cpu().numpy()

    d_bank, d_starts, d_nmix, d_nstd = dev(bank), dev(starts), dev(nmix), dev(nstd)
    got = run(dev(z), 0, dev(cmean), dev(cstd))
    for b in range(B):
        f, ev = DO.mix_samples(bank[None], events, list(starts[b, :nmix[b]]), crop)
        if nstd[b] > 0:
            f = DO.add_noise(f, float(nstd[b]), z[b][None])
        ref = DO.transform_complex(f, cmean, cstd, mel)[0]
        np.testing.assert_allclose(got[b], ref, atol=2e-3, err_msg=f"sample {b}")    # dB
    # in-kernel generator == its numpy restatement (same draws, same result)
    ctr = np.arange(B * crop * bins, dtype=np.uint64)
    zc = DO.counter_normal(77, ctr).reshape(B, crop, bins)
    a = run(None, 77, None, None)
    b_ = run(dev(zc), 0, None, None)
    np.testing.assert_allclose(a, b_, atol=2e-3)
    # a crop that would leave the bank is refused on the host, nothing is launched
    bad = starts.copy()
    bad[2, 1] = frames - crop + 1
    rc = L.lib().sed_complex_augment_logmel(
        L.ptr(d_bank), frames, bad.ctypes.data, nmix.ctypes.data, L.ptr(d_starts), L.ptr(d_nmix), L.ptr(d_nstd), None,
        ctypes.c_ulonglong(0), None, None, L.ptr(fe.melT), L.ptr(fe.mel_lo), L.ptr(fe.mel_hi), L.ptr(out), B, crop, bins,
        64, None)
    assert rc != 0 and b"outside" in L.lib().sed_last_error()


def _write_dataset(root, mode, cfg, rng, n=4, frames=(70, 64, 90, 75)):
    d = os.path.join(root, f"{mode}-features_and_labels")
    os.makedirs(d, exist_ok=True)
    allf = []
    for i in range(n):
        T = frames[i]
        if mode == "logMel":
            f = (10 * rng.standard_normal((1, T, cfg.mel_bins)) - 30).astype(np.float32)
        else:
            f = _complex_bank(rng, T, cfg.bins)[None] * np.float32(5.0)
        allf.append(f)
        with open(os.path.join(d, f"rec{i}_{mode}_features_and_labels.pkl"), "wb") as fh:
            pickle.dump({"features": f, "start_times": [3.0 + i, 15.0], "end_times": [4.0 + i, 16.5]}, fh)
    cat = np.concatenate(allf, axis=1)
    ms = os.path.join(root, f"{mode}-features_mean_std.pkl")
    with open(ms, "wb") as fh:
        pickle.dump({"mean": np.mean(cat, axis=(0, 1)), "std": np.std(cat, axis=(0, 1))}, fh)
    return d, ms


@pytest.mark.parametrize("mode", ["logMel", "Complex"])
def test_spectogram_dataset_end_to_end(env, tmp_path, mode):
    L, sc, pp = env
    dsm = importlib.import_module(PKG + ".dataset.spectogram.spectograms_dataset")
    cfg = sc.SpectogramConfig(3000, 1000, 1000, 1024)          # fps 3 -> crop 30 frames (like REF-NATIVE), 513 bins
    rng = np.random.default_rng(2)
    d, ms = _write_dataset(str(tmp_path), mode, cfg, rng)
    np.random.seed(0)
    ds = dsm.SpectogramDataset(d, ms, val_descriptor="rec3", balance_classes=False, augment_data=False,
                               preprocessed_mode=mode, cfg=cfg)
    assert len(ds) == (70 - 30) + (64 - 30) + (90 - 30)
    x, y = ds[5]
    assert x.is_cuda and x.shape == (1, 30, 64) and x.dtype == torch.float32
    assert y.shape == (30, 1) and y.dtype == torch.float64
    # oracle on the same crop
    files = sorted(p for p in os.listdir(d) if "rec3" not in p)
    order = files          # (the dataset walks the directory sorted: the same order on every data-parallel rank)
    feats = np.concatenate([pickle.load(open(os.path.join(d, p), "rb"))["features"] for p in order], axis=1)
    s = int(ds.train_start_indices[5])
    msd = pickle.load(open(ms, "rb"))
    ocfg = FO.FrontEndConfig(cfg.working_sample_rate, cfg.frame_size, cfg.hop_size, cfg.NFFT)
    if mode == "logMel":
        ref = DO.transform_logmel(feats[:, s:s + 30], msd["mean"], msd["std"])
        np.testing.assert_allclose(x.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    else:
        ref = DO.transform_complex(feats[:, s:s + 30], msd["mean"], msd["std"], FO.mel_filter_bank_matrix(ocfg))
        np.testing.assert_allclose(x.cpu().numpy(), ref, atol=2e-3)
    assert np.array_equal(y.cpu().numpy(), ds.train_event_matrix[s:s + 30])
    assert len(files) == 3
    # the batch loader walks the start table in order, one launch per batch, short tail kept
    loader = dsm.DeviceBatchLoader(ds, 32)
    batches = list(loader)
    assert len(batches) == len(loader) == (len(ds) + 31) // 32
    assert batches[0][0].shape == (32, 1, 30, 64) and batches[-1][0].shape[0] == len(ds) - 32 * (len(batches) - 1)
    np.testing.assert_allclose(batches[0][0][5].cpu().numpy(), x.cpu().numpy(), atol=0)
    # two ranks see disjoint slices of every global step
    r0 = next(iter(dsm.DeviceBatchLoader(ds, 8, rank=0, world_size=2)))
    r1 = next(iter(dsm.DeviceBatchLoader(ds, 8, rank=1, world_size=2)))
    np.testing.assert_allclose(r1[0][0].cpu().numpy(), batches[0][0][8].cpu().numpy(), atol=0)
    np.testing.assert_allclose(r0[0][7].cpu().numpy(), batches[0][0][7].cpu().numpy(), atol=0)
    # validation sampler: whole recording, batch 1
    v = list(ds.get_validation_sampler(3))
    assert len(v) == 1 and v[0][0].shape == (1, 1, 75, 64) and v[0][1].shape == (1, 75, 1)
    assert v[0][2].startswith("rec3")
    if mode == "Complex":
        np.random.seed(1)
        aug = dsm.SpectogramDataset(d, ms, val_descriptor="rec3", augment_data=True, preprocessed_mode=mode, cfg=cfg)
        xa, ya = aug.device_batch(list(range(64)))
        assert torch.isfinite(xa).all() and xa.shape == (64, 1, 30, 64)
        assert ya.shape == (64, 30, 1) and float(ya.max()) <= 1.0
    # train() + device-side eval() run on it
    tr = importlib.import_module(PKG + ".train")
    sed = importlib.import_module(PKG)
    model = sed.Cnn_AvgPooling(1, [(8, 2), (16, 2), (16, 2), (16, 1)])
    trainer = tr.train(model, loader, sed.WeightedBCE(5, True), num_steps=6, lr=1e-3, log_freq=3,
                       outputs_dir=str(tmp_path / "out"), device="cuda")
    assert os.path.exists(tmp_path / "out" / "checkpoints" / "iteration_6.pth")
    losses, rs, ps, aps = tr.eval(model, loader, sed.WeightedBCE(5, True), str(tmp_path / "out"), 6, "cuda", 3)
    assert len(losses) == 1 and rs[0].shape == (21,) and ps[0].shape == (21,) and np.isfinite(aps[0])
    assert trainer is not None


def test_infer_cli_on_a_wav(env, tmp_path):
    from scipy.io import wavfile
    sed = importlib.import_module(PKG)
    infer = importlib.import_module(PKG + ".infer")
    sr = 48000
    rng = np.random.default_rng(0)
    wav = (0.05 * rng.standard_normal(sr * 20)).astype(np.float32)
    p = str(tmp_path / "clip.wav")
    wavfile.write(p, sr, wav)
    model = sed.Cnn_AvgPooling(1, [(32, 2), (64, 2), (128, 2), (128, 1)])
    ck = str(tmp_path / "m.pth")
    torch.save({"iterations": 0, "model": model.state_dict()}, ck)
    infer.main([p, "--ckpt", ck, "--outputs_dir", str(tmp_path / "inf"), "--precision", "fp32"])
    z = np.load(tmp_path / "inf" / "clip.npz")
    T = 1 + (sr * 20) // 15840
    assert z["probabilities"].shape == (8 * (T // 8), 1)
    assert ((z["probabilities"] > 0) & (z["probabilities"] < 1)).all()
    assert np.array_equal(z["decisions"], z["probabilities"] > 0.5)
    assert np.array_equal(infer.onset_frames([0, 1, 1, 0, 1]), [1, 4]) and np.array_equal(infer.onset_frames([1, 0]), [0])
