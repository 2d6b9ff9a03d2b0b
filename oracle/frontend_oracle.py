"""CPU oracle for the log-mel front-end of the SED hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.

Restates, in numpy, what the reference computes at
    MEL_FILTER_BANK_MATRIX                /root/reference/dataset/spectogram/preprocess.py:13-18
    multichannel_stft                     /root/reference/dataset/spectogram/preprocess.py:21-36
    multichannel_complex_to_log_mel       /root/reference/dataset/spectogram/preprocess.py:39-45
    calculate_scalar_of_tensor            /root/reference/dataset/spectogram/preprocess.py:48-57
    SpectogramDataset.transform           /root/reference/dataset/spectogram/spectograms_dataset.py:104-110

PARITY UNPINNED at the librosa boundary.  The arithmetic of these rows lives in the third-party
package `librosa`, which is not vendored in /root/reference, not pinned by it (README.md:11-13 only
names it; the keyword usage `librosa.filters.mel(sr=...)`, `librosa.core.stft`,
`librosa.core.power_to_db` implies 0.8-0.10) and not installed in the build image, and the
reference has no tests or golden vectors.  This file therefore restates librosa's *published*
algorithm for exactly the arguments the reference's call sites pass:

  librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with defaults htk=False, norm='slaney':
      fftfreqs = linspace(0, sr/2, 1+n_fft//2); n_mels+2 band edges equally spaced on the Slaney
      mel scale (linear 200/3 Hz per mel below 1 kHz, log above with step ln(6.4)/27);
      triangular weights max(0, min(lower, upper)); each filter scaled by 2/(f[i+2]-f[i]); float32.
  librosa.core.stft(y, n_fft, hop_length, win_length, window=<array>, center=True,
                    pad_mode='reflect', dtype=complex64):
      window zero-padded to n_fft, centred ((n_fft-win)//2 on the left); y reflect-padded by
      n_fft//2 each side (edge sample not repeated); frame t = padded[t*hop : t*hop+n_fft];
      T = 1 + len(y)//hop; rFFT of window*frame; result (bins, T) -> reference transposes.
  librosa.core.power_to_db(S, ref=1.0, amin=1e-10, top_db=None) = 10*log10(maximum(amin, S)).

Round 5: cross-checked against a THIRD-PARTY port of the same librosa functions that is installed in the image,
`transformers.audio_utils` (`mel_filter_bank(norm="slaney", mel_scale="slaney")`, `spectrogram(center=True,
pad_mode="reflect", power=2.0, log_mel="dB")`): the filter bank agrees to 5e-8 of its scale, the whole
waveform -> log-mel chain to 5e-7 dB, for BENCH and REF-NATIVE (tests/test_frontend_oracle.py).  That is not the
reference's own librosa, so the header keeps saying "unpinned" -- but no code is shared with that port.

It is anchored on what the reference itself offers: its numpy-only STFT variant
(Classical_methods/train_svm_detector.py:65-68: `frames *= np.hanning(n)`, `np.fft.rfft(frames,
NFFT)`, same log-mel function) which `tests/test_frontend_oracle.py` reproduces, plus known-answer
tests (bin-centred sinusoid, Parseval, unit-area Slaney filters, -100 dB floor).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import numpy as np


@dataclass(frozen=True)
class FrontEndConfig:
    """Runtime counterpart of dataset/common_config.py:2-8 + spectogram_configs.py:5-10."""
    sample_rate: int
    frame_size: int        # window length
    hop_size: int
    nfft: int
    mel_bins: int = 64
    mel_min_freq: float = 20.0
    mel_max_freq: float = None  # default sample_rate//2

    @property
    def fmax(self) -> float:
        return float(self.sample_rate // 2) if self.mel_max_freq is None else float(self.mel_max_freq)

    @property
    def bins(self) -> int:
        return self.nfft // 2 + 1

    def num_frames(self, samples: int) -> int:
        return 1 + samples // self.hop_size


def ref_native_config() -> FrontEndConfig:
    """The committed constants: 48 kHz, time_margin .33 -> frame 31680, hop 15840, NFFT 32768."""
    sr = 48000
    frame = int(sr * 0.33 * 2)
    nfft = 2 ** int(np.ceil(np.log2(frame)))
    return FrontEndConfig(sr, frame, frame // 2, nfft)


def bench_config() -> FrontEndConfig:
    """SURVEY D3 'BENCH': 32 kHz, win = nfft = 1024, hop 320 -> 6001 frames per 60 s."""
    return FrontEndConfig(32000, 1024, 320, 1024)


# ----------------------------------------------------------------------------------------------
# mel filterbank (librosa.filters.mel, Slaney scale + slaney norm)
# ----------------------------------------------------------------------------------------------
_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP          # 15
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    mel = f / _F_SP
    log_t = f >= _MIN_LOG_HZ
    return np.where(log_t, _MIN_LOG_MEL + np.log(np.maximum(f, 1e-300) / _MIN_LOG_HZ) / _LOGSTEP, mel)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f = _F_SP * m
    log_t = m >= _MIN_LOG_MEL
    return np.where(log_t, _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL)), f)


def mel_filter_bank_matrix(cfg: FrontEndConfig) -> np.ndarray:
    """MEL_FILTER_BANK_MATRIX (preprocess.py:13-18): (nfft/2+1, mel_bins) float32 (the transpose)."""
    n_mels = cfg.mel_bins
    fftfreqs = np.linspace(0.0, cfg.sample_rate / 2.0, cfg.bins)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(cfg.mel_min_freq), hz_to_mel(cfg.fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, cfg.bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32).T.copy()


# ----------------------------------------------------------------------------------------------
# STFT
# ----------------------------------------------------------------------------------------------
def padded_window(cfg: FrontEndConfig) -> np.ndarray:
    """np.hanning(frame_size) (symmetric), zero-padded and centred to nfft (librosa pad_center)."""
    w = np.hanning(cfg.frame_size)
    lpad = (cfg.nfft - cfg.frame_size) // 2
    out = np.zeros(cfg.nfft, dtype=np.float64)
    out[lpad:lpad + cfg.frame_size] = w
    return out


def stft_channel(y: np.ndarray, cfg: FrontEndConfig, dtype=np.complex64) -> np.ndarray:
    """One channel -> (T, bins) complex (already transposed like preprocess.py:33)."""
    y = np.asarray(y)
    ypad = np.pad(y, cfg.nfft // 2, mode="reflect")
    T = cfg.num_frames(len(y))
    win = padded_window(cfg).astype(y.dtype if np.issubdtype(y.dtype, np.floating) else np.float64)
    out = np.empty((T, cfg.bins), dtype=dtype)
    chunk = max(1, (1 << 24) // cfg.nfft)
    for t0 in range(0, T, chunk):
        t1 = min(T, t0 + chunk)
        idx = (np.arange(t0, t1) * cfg.hop_size)[:, None] + np.arange(cfg.nfft)[None, :]
        out[t0:t1] = np.fft.rfft(ypad[idx] * win[None, :], axis=1).astype(dtype)
    return out


def multichannel_stft(sig: np.ndarray, cfg: FrontEndConfig) -> np.ndarray:
    """(samples, channels) -> (channels, T, bins) complex64 (preprocess.py:21-36)."""
    return np.array([stft_channel(sig[:, c], cfg) for c in range(sig.shape[1])])


def multichannel_complex_to_log_mel(X: np.ndarray, mel: np.ndarray) -> np.ndarray:
    """abs()**2 -> dot(MEL) -> 10*log10(max(1e-10, .)) -> float32 (preprocess.py:39-45). Any
    leading dims (used 2-D at Classical_methods/train_svm_detector.py:68)."""
    power = np.abs(X) ** 2
    melspec = np.dot(power, mel)
    return (10.0 * np.log10(np.maximum(1e-10, melspec))).astype(np.float32)


def calculate_scalar_of_tensor(x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Per-mel-bin mean / population std over (channels, frames) (preprocess.py:48-57)."""
    axis = 0 if x.ndim == 2 else (0, 1)
    return np.mean(x, axis=axis), np.std(x, axis=axis)


def transform(x: np.ndarray, mean: np.ndarray, std: np.ndarray) -> np.ndarray:
    """SpectogramDataset.transform, logMel mode (spectograms_dataset.py:104-108)."""
    return (x - mean) / std


def log_mel_from_waveform(sig: np.ndarray, cfg: FrontEndConfig, mean=None, std=None,
                          precision: str = "ref") -> np.ndarray:
    """Waveform (samples, channels) -> (channels, T, mel) float32, optionally z-scored.

    precision='ref': complex64 STFT, float32 power/mel as the reference pipeline does.
    precision='f64': everything in float64 (the 'truth' used to bound both)."""
    mel = mel_filter_bank_matrix(cfg)
    if precision == "ref":
        X = multichannel_stft(sig, cfg)
        lm = multichannel_complex_to_log_mel(X, mel)
    else:
        X = np.array([stft_channel(sig[:, c].astype(np.float64), cfg, np.complex128)
                      for c in range(sig.shape[1])])
        lm = 10.0 * np.log10(np.maximum(1e-10, np.dot(np.abs(X) ** 2, mel.astype(np.float64))))
    if mean is not None:
        lm = transform(lm, mean, std)
    return lm


def svm_variant_log_mel(frames: np.ndarray, cfg: FrontEndConfig) -> np.ndarray:
    """The reference's own numpy-only pipeline (Classical_methods/train_svm_detector.py:65-68):
    frames (N, frame_size) * np.hanning -> np.fft.rfft(frames, NFFT) (right zero pad) -> log-mel."""
    fr = frames * np.hanning(frames.shape[1])
    return multichannel_complex_to_log_mel(np.fft.rfft(fr, cfg.nfft), mel_filter_bank_matrix(cfg))
