"""Known-answer tests for oracle/frontend_oracle.py (librosa semantics; parity unpinned at the
librosa boundary -- see that file's header) and the cross-check against the reference's own
numpy-only STFT variant (Classical_methods/train_svm_detector.py:65-68)."""
import numpy as np
import pytest

from oracle import frontend_oracle as FO


def test_configs_match_reference_constants():
    c = FO.ref_native_config()    # dataset/common_config.py:2-8, spectogram_configs.py:5-10
    assert (c.sample_rate, c.frame_size, c.hop_size, c.nfft, c.mel_bins) == (48000, 31680, 15840, 32768, 64)
    assert c.bins == 16385 and c.num_frames(60 * 48000) == 182
    b = FO.bench_config()
    assert b.num_frames(60 * 32000) == 6001 and b.bins == 513


@pytest.mark.parametrize("cfg", [FO.bench_config(), FO.ref_native_config()])
def test_mel_filters_slaney_unit_area_and_shape(cfg):
    M = FO.mel_filter_bank_matrix(cfg)
    assert M.shape == (cfg.bins, 64) and M.dtype == np.float32
    assert (M >= 0).all()
    df = cfg.sample_rate / cfg.nfft
    area = M.astype(np.float64).sum(axis=0) * df
    # slaney norm: each triangle has unit area (discretisation error shrinks with nfft)
    tol = 0.35 if cfg.nfft == 1024 else 0.02
    assert np.all(np.abs(area[4:] - 1.0) < tol)
    # centre frequencies increase, lowest filter starts at fmin
    peaks = M.argmax(axis=0)
    assert np.all(np.diff(peaks) >= 0)
    assert M[: int(np.floor(cfg.mel_min_freq / df)), :].sum() == 0


def test_mel_scale_known_values():
    assert FO.hz_to_mel(1000.0) == pytest.approx(15.0)
    assert FO.hz_to_mel(200.0) == pytest.approx(3.0)
    assert FO.mel_to_hz(FO.hz_to_mel(6400.0)) == pytest.approx(6400.0)
    assert FO.hz_to_mel(6400.0) == pytest.approx(15.0 + 27.0)   # log step defined by 6.4 kHz = 27 steps


def test_window_is_symmetric_hann_centred():
    cfg = FO.ref_native_config()
    w = FO.padded_window(cfg)
    l = (cfg.nfft - cfg.frame_size) // 2
    assert w[:l].sum() == 0 and w[l + cfg.frame_size:].sum() == 0
    assert w[l] == 0 and w[l + cfg.frame_size - 1] == pytest.approx(0, abs=1e-15)
    np.testing.assert_allclose(w[l:l + cfg.frame_size], w[l:l + cfg.frame_size][::-1], atol=1e-15)


def test_stft_frame_count_reflect_and_sinusoid_peak():
    cfg = FO.bench_config()
    n = 32000
    k0 = 100                                        # bin-centred tone
    t = np.arange(n)
    y = np.cos(2 * np.pi * k0 * t / cfg.nfft)
    X = FO.stft_channel(y, cfg)
    assert X.shape == (1 + n // cfg.hop_size, cfg.bins) and X.dtype == np.complex64
    mid = np.abs(X[10:-10])
    assert np.all(mid.argmax(axis=1) == k0)
    # Hann main lobe: |X[k0]| = sum(w)/2, neighbours half of that
    assert mid[:, k0].mean() == pytest.approx(np.hanning(cfg.frame_size).sum() / 2, rel=1e-3)
    assert (mid[:, k0 + 1] / mid[:, k0]).mean() == pytest.approx(0.5, rel=2e-2)
    # frame 0 sees the reflect padding: padded[0:nfft] = y[512:0:-1] ++ y[0:512]
    ypad = np.pad(y, cfg.nfft // 2, mode="reflect")
    assert ypad[0] == y[cfg.nfft // 2] and ypad[cfg.nfft // 2] == y[0] and ypad[cfg.nfft // 2 - 1] == y[1]
    np.testing.assert_allclose(X[0], np.fft.rfft(ypad[: cfg.nfft] * FO.padded_window(cfg)), rtol=1e-4, atol=1e-3)


def test_parseval_per_frame():
    cfg = FO.bench_config()
    rng = np.random.default_rng(0)
    y = rng.standard_normal(8000)
    X = FO.stft_channel(y, cfg, dtype=np.complex128)
    ypad = np.pad(y, cfg.nfft // 2, mode="reflect")
    fr = ypad[5 * cfg.hop_size: 5 * cfg.hop_size + cfg.nfft] * FO.padded_window(cfg)
    p = np.abs(X[5]) ** 2
    total = p[0] + p[-1] + 2 * p[1:-1].sum()
    assert total / cfg.nfft == pytest.approx((fr ** 2).sum(), rel=1e-10)


def test_log_floor_and_dtype():
    cfg = FO.bench_config()
    sil = np.zeros((4000, 1))
    lm = FO.log_mel_from_waveform(sil, cfg)
    assert lm.dtype == np.float32 and lm.shape == (1, 13, 64)
    # 10*log10(float32(1e-10)): the float32 rounding of amin gives -100.00001, as in the reference
    assert np.all(np.abs(lm + 100.0) < 2e-5) and np.all(lm == lm[0, 0, 0])


def test_matches_reference_numpy_variant_up_to_window_shift():
    """train_svm_detector.py:65-68 zero-pads on the RIGHT (np.fft.rfft(frames, NFFT)); librosa
    centres the window.  A circular shift changes phases only -> identical magnitudes/log-mel."""
    cfg = FO.ref_native_config()
    rng = np.random.default_rng(1)
    y = rng.standard_normal(3 * cfg.hop_size + 5) * 0.1
    ypad = np.pad(y, cfg.nfft // 2, mode="reflect")
    l = (cfg.nfft - cfg.frame_size) // 2
    T = cfg.num_frames(len(y))
    frames = np.stack([ypad[t * cfg.hop_size + l: t * cfg.hop_size + l + cfg.frame_size] for t in range(T)])
    a = FO.svm_variant_log_mel(frames, cfg)
    b = FO.log_mel_from_waveform(y[:, None], cfg)[0]
    np.testing.assert_allclose(a, b, atol=2e-3)
    c = FO.log_mel_from_waveform(y[:, None], cfg, precision="f64")[0]
    np.testing.assert_allclose(b, c, atol=2e-3)


def test_normalisation_stats():
    x = np.random.default_rng(2).standard_normal((2, 50, 64)).astype(np.float32) * 3 + 1
    m, s = FO.calculate_scalar_of_tensor(x)
    assert m.shape == (64,) and s.shape == (64,)
    z = FO.transform(x, m, s)
    np.testing.assert_allclose(z.mean(axis=(0, 1)), 0, atol=1e-5)
    np.testing.assert_allclose(z.std(axis=(0, 1)), 1, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------------------------
# Independent implementations (round-3 verdict, task 7a).  The oracle stays "parity unpinned" at the librosa boundary (librosa
# is not in the image and /root/reference holds no fixtures), but torch.stft and scipy.signal are third-party STFTs written by
# other people: agreement rules out a mistake SHARED by the oracle and the HIP kernel (both were written here).
# Call-site semantics restated: /root/reference/dataset/spectogram/preprocess.py:25-33 -- librosa.core.stft(n_fft, hop_length,
# win_length, window=np.hanning(win), center=True, pad_mode='reflect').
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,n", [(FO.bench_config(), 32000 + 123), (FO.ref_native_config(), 3 * 15840 + 77)])
def test_stft_matches_torch_stft(cfg, n):
    import torch
    rng = np.random.default_rng(3)
    y = rng.standard_normal(n) * 0.1
    X = FO.stft_channel(y, cfg, dtype=np.complex128)
    win = torch.from_numpy(np.hanning(cfg.frame_size))        # symmetric Hann; torch centres a short window in n_fft itself
    S = torch.stft(torch.from_numpy(y), n_fft=cfg.nfft, hop_length=cfg.hop_size, win_length=cfg.frame_size, window=win,
                   center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    S = S.numpy().T                                            # (bins, T) -> (T, bins)
    assert S.shape == X.shape == (1 + n // cfg.hop_size, cfg.bins)
    scale = np.abs(S).max()
    assert np.abs(X - S).max() / scale < 1e-12


def test_stft_matches_scipy_short_time_fft():
    """scipy.signal.ShortTimeFFT with the same padded window, hop and FFT length; scipy zero-pads at the borders, so only the
    frames that do not touch the reflect padding are compared (the padding itself is checked against np.pad / torch above)."""
    from scipy.signal import ShortTimeFFT
    cfg = FO.bench_config()
    rng = np.random.default_rng(4)
    n = 20000
    y = rng.standard_normal(n) * 0.1
    X = FO.stft_channel(y, cfg, dtype=np.complex128)
    sft = ShortTimeFFT(FO.padded_window(cfg), hop=cfg.hop_size, fs=cfg.sample_rate, mfft=cfg.nfft, fft_mode="onesided",
                       phase_shift=None)
    S = sft.stft(y)                                            # (bins, slices); slice p is centred at sample p * hop
    p0 = -sft.p_min                                            # index of the slice centred at sample 0
    T = X.shape[0]
    inner = [t for t in range(T) if t * cfg.hop_size - cfg.nfft // 2 >= 0 and t * cfg.hop_size + cfg.nfft // 2 <= n]
    assert len(inner) > 40
    got = np.stack([S[:, p0 + t] for t in inner])
    # phase_shift=None references the phase to the window START, numpy's rfft of the frame does too: direct comparison
    assert np.abs(got - X[inner]).max() / np.abs(X).max() < 1e-12


def test_log_mel_chain_matches_float64_scipy_chain():
    """|STFT|^2 -> mel -> 10 log10 recomputed with scipy's STFT and a float64 mel product on the inner frames."""
    from scipy.signal import ShortTimeFFT
    cfg = FO.bench_config()
    rng = np.random.default_rng(5)
    n = 16000
    y = rng.standard_normal(n) * 0.05 + 0.3 * np.sin(2 * np.pi * 1000 * np.arange(n) / cfg.sample_rate)
    lm = FO.log_mel_from_waveform(y[:, None], cfg)[0]
    sft = ShortTimeFFT(FO.padded_window(cfg), hop=cfg.hop_size, fs=cfg.sample_rate, mfft=cfg.nfft, fft_mode="onesided",
                       phase_shift=None)
    S = sft.stft(y)
    p0 = -sft.p_min
    M = FO.mel_filter_bank_matrix(cfg).astype(np.float64)
    inner = [t for t in range(lm.shape[0]) if t * cfg.hop_size - cfg.nfft // 2 >= 0 and t * cfg.hop_size + cfg.nfft // 2 <= n]
    ref = 10.0 * np.log10(np.maximum(1e-10, (np.abs(np.stack([S[:, p0 + t] for t in inner])) ** 2) @ M))
    np.testing.assert_allclose(lm[inner], ref, atol=2e-3)


@pytest.mark.parametrize("cfg", [FO.bench_config(), FO.ref_native_config()])
def test_mel_filter_bank_against_a_second_derivation(cfg):
    """MEL_FILTER_BANK_MATRIX (/root/reference/dataset/spectogram/preprocess.py:13-18 = librosa.filters.mel, Slaney scale, Slaney norm) derived a
    second time WITHOUT any code of oracle/frontend_oracle.py: the published scale written as a scalar function (linear 200/3 Hz per mel below
    1 kHz, 27 log steps per factor 6.4 above), the n_mels + 2 band edges found by root finding (scipy.optimize.brentq) on that function at
    equally spaced mel values, every weight by direct evaluation of the band's triangle at the FFT bin centre, times 2 / band width.  Does not
    pin the oracle to librosa (absent from this image) -- it rules out an indexing / transposition / off-by-one slip in the one piece of the
    front-end that torch.stft and scipy.signal do not cover (task 6a)."""
    import math

    from scipy.optimize import brentq

    def mel_of(f):
        return f * 3.0 / 200.0 if f < 1000.0 else 15.0 + 27.0 * math.log(f / 1000.0) / math.log(6.4)

    n = cfg.mel_bins
    m_lo, m_hi = mel_of(cfg.mel_min_freq), mel_of(cfg.fmax)
    edges = []
    for k in range(n + 2):
        target = m_lo + (m_hi - m_lo) * k / (n + 1)
        edges.append(brentq(lambda f: mel_of(f) - target, 0.0, cfg.sample_rate, xtol=1e-10, rtol=1e-14))
    ref = np.zeros((cfg.bins, n))
    for j in range(cfg.bins):
        f = j * cfg.sample_rate / cfg.nfft                  # centre of rFFT bin j (bins = nfft/2 + 1: the last one is Nyquist)
        for i in range(n):
            lo, mid, hi = edges[i], edges[i + 1], edges[i + 2]
            if lo < f < hi:
                tri = (f - lo) / (mid - lo) if f <= mid else (hi - f) / (hi - mid)
                ref[j, i] = tri * 2.0 / (hi - lo)
    M = FO.mel_filter_bank_matrix(cfg).astype(np.float64)
    assert M.shape == ref.shape
    assert np.abs(M - ref).max() <= 2e-7 * ref.max() + 1e-12, float(np.abs(M - ref).max() / ref.max())
    assert ((M > 0) == (ref > 1e-12 * ref.max())).mean() > 0.9999       # the same support (up to a bin that sits on a band edge)
    # every bin between the first and the last edge is covered by at least one band (no hole from an off-by-one in the edge list)
    covered = (ref > 0).any(axis=1)
    f_bins = np.arange(cfg.bins) * cfg.sample_rate / cfg.nfft
    inside = (f_bins > edges[0]) & (f_bins < edges[-1])
    assert covered[inside].all()


# ----------------------------------------------------------------------------------------------
# Round 5: a THIRD-PARTY implementation of librosa's semantics that is installed in the image.  `transformers.audio_utils`
# (Hugging Face) ships `mel_filter_bank(norm="slaney", mel_scale="slaney")` -- documented as equivalent to `librosa.filters.mel` --
# and `spectrogram(center=True, pad_mode="reflect", power=2.0, log_mel="dB")`, their port of `librosa.stft` -> mel -> `power_to_db`.
# It shares no code with oracle/frontend_oracle.py or with the kernels.  It does not lift "parity unpinned" (it is not the
# reference's librosa, and the reference holds no fixtures), but it rules out a mistake common to the builder's restatement and
# its kernels for EVERY front-end row (a1 mel filter bank, a2 STFT framing / padding / window, a3 power -> mel -> dB clamp),
# including the one piece the scipy / torch.stft cross-checks above do not reach: the Slaney filter bank.
# ----------------------------------------------------------------------------------------------
au = pytest.importorskip("transformers.audio_utils")


@pytest.mark.parametrize("cfg", [FO.bench_config(), FO.ref_native_config()])
def test_mel_filter_bank_matches_transformers_port_of_librosa(cfg):
    ours = FO.mel_filter_bank_matrix(cfg).astype(np.float64)                     # (bins, n_mels)
    theirs = au.mel_filter_bank(num_frequency_bins=cfg.bins, num_mel_filters=cfg.mel_bins, min_frequency=cfg.mel_min_freq,
                                max_frequency=cfg.fmax, sampling_rate=cfg.sample_rate, norm="slaney", mel_scale="slaney")
    assert theirs.shape == ours.shape
    scale = np.abs(theirs).max()
    assert np.abs(ours - theirs).max() < 2e-7 * scale            # (ours is rounded to float32 like librosa's; measured 4-5e-8)
    assert ((ours > 0) == (theirs > 1e-12 * scale)).mean() > 0.9999      # the same support (band edges on the same bins)


@pytest.mark.parametrize("cfg,seconds", [(FO.bench_config(), 1.5), (FO.ref_native_config(), 4.0)])
def test_log_mel_chain_matches_transformers_port_of_librosa(cfg, seconds):
    rng = np.random.default_rng(11)
    n = int(seconds * cfg.sample_rate)
    t = np.arange(n) / cfg.sample_rate
    y = (0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.05 * rng.standard_normal(n) + 0.2 * np.sin(2 * np.pi * 5000.0 * t) * (t > seconds / 2)).astype(np.float64)
    y[: n // 10] *= 1e-4                                          # a quiet stretch: exercises the low end of the dB range
    ours = FO.log_mel_from_waveform(y[:, None], cfg, precision="f64")[0]         # (T, n_mels)
    mel = au.mel_filter_bank(num_frequency_bins=cfg.bins, num_mel_filters=cfg.mel_bins, min_frequency=cfg.mel_min_freq,
                             max_frequency=cfg.fmax, sampling_rate=cfg.sample_rate, norm="slaney", mel_scale="slaney")
    theirs = au.spectrogram(y, window=FO.padded_window(cfg), frame_length=cfg.nfft, hop_length=cfg.hop_size, fft_length=cfg.nfft,
                            power=2.0, center=True, pad_mode="reflect", onesided=True, mel_filters=mel, mel_floor=1e-10,
                            log_mel="dB", reference=1.0, min_value=1e-10, dtype=np.float64).T
    assert theirs.shape == ours.shape == (cfg.num_frames(n), cfg.mel_bins)
    assert np.abs(ours - theirs).max() < 1e-5, np.abs(ours - theirs).max()       # dB; (ours: float32 filter weights, theirs float64; measured 3-5e-7)
    # and the reference-precision path (complex64 STFT, float32 mel) stays within the same band
    ref32 = FO.log_mel_from_waveform(y[:, None].astype(np.float32), cfg, precision="ref")[0]
    assert np.abs(ref32 - theirs).max() < 1e-3                   # (measured 1-1.5e-5 dB)
