"""Log-mel front-end on the MI355X vs the numpy oracle (librosa semantics; see oracle header for
the pinning status).  Tolerances: STFT bins within 2e-5 of the frame's largest magnitude (fp32 LDS
FFT vs float64 FFT rounded to complex64); log-mel within 2e-3 dB."""
import importlib

import numpy as np
import pytest
import torch

from oracle import frontend_oracle as FO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available()
    pp = importlib.import_module("soundeventdetection-pytorch_amd.dataset.spectogram.preprocess")
    sc = importlib.import_module("soundeventdetection-pytorch_amd.dataset.spectogram.spectogram_configs")
    return pp, sc


def signal(n, sr, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    y = 0.1 * rng.standard_normal(n)
    y += 0.5 * np.sin(2 * np.pi * 1234.5 * t) * np.hanning(n)
    y[n // 3: n // 3 + 200] += 0.8 * rng.standard_normal(200)
    return np.clip(y, -1, 1)


def ocfg(c):
    return FO.FrontEndConfig(c.working_sample_rate, c.frame_size, c.hop_size, c.NFFT)


@pytest.mark.parametrize("which,seconds", [("BENCH", 2.0), ("REF_NATIVE", 4.0)])
def test_stft_and_logmel_match_oracle(mods, which, seconds):
    pp, sc = mods
    c = getattr(sc, which)
    n = int(seconds * c.working_sample_rate) + 17
    waves = np.stack([signal(n, c.working_sample_rate, s) for s in (0, 1)])
    fe = pp.LogMelFrontEnd(c, "cuda")
    spec = fe.stft(waves.astype(np.float32)).cpu().numpy()
    ref = np.stack([FO.stft_channel(w.astype(np.float32).astype(np.float64), ocfg(c), np.complex128) for w in waves])
    assert spec.shape == ref.shape == (2, 1 + n // c.hop_size, c.NFFT // 2 + 1)
    scale = np.abs(ref).max(axis=2, keepdims=True)
    assert (np.abs(spec - ref) / scale).max() < 2e-5
    lm = fe(waves.astype(np.float32)).cpu().numpy()
    lm_ref = np.stack([FO.log_mel_from_waveform(w.astype(np.float32)[:, None], ocfg(c))[0] for w in waves])
    assert lm.shape == (2, 1, ref.shape[1], 64)
    np.testing.assert_allclose(lm[:, 0], lm_ref, atol=2e-3)
    # complex -> log-mel entry point ("Complex" preprocessing mode) on the oracle's own spectrogram
    X = FO.multichannel_stft(waves.T.astype(np.float32), ocfg(c))
    lm2 = fe.complex_to_log_mel(X).cpu().numpy()
    np.testing.assert_allclose(lm2, FO.multichannel_complex_to_log_mel(X, FO.mel_filter_bank_matrix(ocfg(c))), atol=2e-4)


@pytest.mark.parametrize("nfft", [4096, 8192, 16384])
def test_large_transform_kernel_other_sizes(mods, nfft):
    """csrc/sed_frontend.hip: frontend_big_kernel (round 6: nfft >= 4096, three radix-2 stages per LDS round trip).  log2(nfft / 2) = 11, 12,
    13 exercise the 3 + 3 + 3 + 2, 3 x 4 and 3 x 4 + 1 stage groupings (the reference's own nfft 32768 = 14 = 3 x 4 + 2 runs in
    test_stft_and_logmel_match_oracle); window shorter than the transform, hop not a divisor of it, a clip that ends inside a frame."""
    pp, sc = mods
    c = sc.SpectogramConfig(44100, nfft - 300, nfft // 3 + 1, nfft)
    n = 3 * nfft + 123
    waves = np.stack([signal(n, c.working_sample_rate, s) for s in (5, 6)])
    fe = pp.LogMelFrontEnd(c, "cuda")
    spec = fe.stft(waves.astype(np.float32)).cpu().numpy()
    ref = np.stack([FO.stft_channel(w.astype(np.float32).astype(np.float64), ocfg(c), np.complex128) for w in waves])
    assert spec.shape == ref.shape == (2, 1 + n // c.hop_size, nfft // 2 + 1)
    scale = np.abs(ref).max(axis=2, keepdims=True)
    assert (np.abs(spec - ref) / scale).max() < 2e-5
    lm = fe(waves.astype(np.float32)).cpu().numpy()
    lm_ref = np.stack([FO.log_mel_from_waveform(w.astype(np.float32)[:, None], ocfg(c))[0] for w in waves])
    np.testing.assert_allclose(lm[:, 0], lm_ref, atol=2e-3)


def test_normalisation_silence_and_short_clip(mods):
    pp, sc = mods
    c = sc.BENCH
    mean = np.linspace(-40, -20, 64).astype(np.float32)
    std = np.linspace(5, 9, 64).astype(np.float32)
    fe = pp.LogMelFrontEnd(c, "cuda", mean=mean, std=std)
    w = signal(c.NFFT // 2 + 5, c.working_sample_rate, 3).astype(np.float32)   # shortest legal clip
    lm = fe(w[None]).cpu().numpy()[0, 0]
    ref = FO.log_mel_from_waveform(w[:, None], ocfg(c), mean, std)[0]
    np.testing.assert_allclose(lm, ref, atol=1e-3)
    sil = fe(np.zeros((1, 4000), np.float32)).cpu().numpy()[0, 0]
    np.testing.assert_allclose(sil, np.broadcast_to((np.float32(-100.0) - mean[None, :]) / std[None, :], sil.shape), atol=1e-4)
    with pytest.raises(RuntimeError):
        fe(np.zeros((1, c.NFFT // 2), np.float32))      # reflect padding impossible


def test_module_level_api_reference_constants(mods):
    pp, sc = mods
    c = sc.REF_NATIVE
    n = 3 * c.hop_size + 11
    sig = np.stack([signal(n, 48000, 7)], axis=1)            # (samples, channels=1)
    X = pp.multichannel_stft(sig)
    assert X.shape == (1, 1 + n // c.hop_size, 16385) and X.dtype == np.complex64
    lm = pp.multichannel_complex_to_log_mel(X)
    assert lm.shape == (1, X.shape[1], 64) and lm.dtype == np.float32
    ref = FO.log_mel_from_waveform(sig.astype(np.float32), FO.ref_native_config())
    np.testing.assert_allclose(lm, ref, atol=2e-3)


def test_full_size_bench_clip_properties(mods):
    """60 s @ 32 kHz: frame count, Parseval of the STFT kernel, time-shift by one hop."""
    pp, sc = mods
    c = sc.BENCH
    n = 60 * c.working_sample_rate
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(2, n, generator=g) * 0.1).clamp_(-1, 1).cuda()
    fe = pp.LogMelFrontEnd(c, "cuda")
    lm = fe(w)
    assert lm.shape == (2, 1, 6001, 64) and torch.isfinite(lm).all()
    spec = fe.stft(w[:, : 50 * c.hop_size])
    p = spec.abs() ** 2
    total = p[..., 0] + p[..., -1] + 2 * p[..., 1:-1].sum(-1)
    # Parseval: sum |X|^2 / nfft == sum (w*x)^2 ; check interior frames on the host
    win = torch.from_numpy(pp.padded_window(c)).cuda()
    fr = w[0, 10 * c.hop_size - 512: 10 * c.hop_size + 512] * win
    assert abs(total[0, 10].item() / c.NFFT - (fr ** 2).sum().item()) / (fr ** 2).sum().item() < 1e-4
    # shifting the waveform by exactly one hop shifts interior frames by one
    lm_s = fe(torch.roll(w, -c.hop_size, dims=1))
    assert torch.allclose(lm[:, :, 5:5000], lm_s[:, :, 4:4999], atol=1e-3)


@pytest.mark.parametrize("n_mels,width,hop,seconds", [(40, 150, 320, 1.3), (64, 30, 128, 0.7), (17, 400, 256, 0.9)])
def test_batched_kernel_other_filter_banks_and_hops(mods, n_mels, width, hop, seconds):
    """The batched nfft-1024 kernel derives its (mel tile, bin range) schedule from the filter bands it is given: banks with wide,
    dense bands (more k-steps per wave than it keeps in registers), fewer than 64 filters (empty tiles), small hops (short batches),
    frame counts that are not a multiple of the batch -- against float64 numpy on the oracle's power spectra."""
    pp, sc = mods
    c = sc.BENCH
    L = importlib.import_module("soundeventdetection-pytorch_amd._lib")
    oc = FO.FrontEndConfig(c.working_sample_rate, c.frame_size, hop, c.NFFT)
    n = (int(seconds * c.working_sample_rate) // 4) * 4
    waves = np.stack([signal(n, c.working_sample_rate, s) for s in (5, 6, 7)]).astype(np.float32)
    rng = np.random.default_rng(n_mels)
    W = np.zeros((n_mels, 513), np.float32)
    lo = np.zeros(n_mels, np.int32)
    hi = np.zeros(n_mels, np.int32)
    for m in range(n_mels):
        a = int(rng.integers(0, 513 - width))
        b = a + int(rng.integers(width // 2, width + 1))
        W[m, a:b] = rng.random(b - a).astype(np.float32) + 0.05
        lo[m], hi[m] = a, b
    fe = pp.LogMelFrontEnd(c, "cuda")
    T = 1 + n // hop
    out = torch.empty(3, T, n_mels, device="cuda")
    w = torch.from_numpy(waves).cuda()
    melT, mlo, mhi = torch.from_numpy(W).cuda(), torch.from_numpy(lo).cuda(), torch.from_numpy(hi).cuda()
    L.check(L.lib().sed_logmel_fwd(L.ptr(w), L.ptr(fe.window), L.ptr(melT), L.ptr(mlo), L.ptr(mhi), None, None, L.ptr(out), L.ptr(fe.ws),
                                   3, n, c.NFFT, hop, n_mels, torch.cuda.current_stream().cuda_stream))
    got = out.cpu().numpy()
    for i in range(3):
        X = FO.stft_channel(waves[i].astype(np.float64), oc, np.complex128)
        ref = 10.0 * np.log10(np.maximum(1e-10, (np.abs(X) ** 2) @ W.astype(np.float64).T))
        assert ref.shape == (T, n_mels)
        np.testing.assert_allclose(got[i], ref, atol=2e-3)


def test_twiddle_table_cache_streams_and_graph_capture(mods):
    """Round 5: the FFT twiddle table is library-owned per (device, nfft) -- built by the first call, reused by later calls on ANY stream,
    and bypassed (workspace path, table launch inside the capture) while a stream is being captured.  The outputs of the three paths are
    bit-identical."""
    pp, sc = mods
    c = sc.BENCH
    n = 3 * c.working_sample_rate
    w = torch.from_numpy(signal(n, c.working_sample_rate, 3).astype(np.float32))[None].repeat(2, 1).cuda()
    fe = pp.LogMelFrontEnd(c, "cuda")
    ref = fe(w).clone()                               # (the table exists from here on at the latest)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        out_side = fe(w).clone()
    side.synchronize()
    assert torch.equal(out_side, ref)
    # capture: no allocation / synchronisation may happen, the kernel takes the table launch into the graph
    static_out = torch.empty_like(ref)
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        fe(w, out=static_out)                         # warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            fe(w, out=static_out)
    static_out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, ref)
