"""Counterpart of /root/reference/dataset/spectogram/preprocess.py:13-57 with the arithmetic in
libsed_hip.so: framing + window + LDS FFT + power + mel + log (+ z-score) on the MI355X.

    MEL_FILTER_BANK_MATRIX            (bins, mel_bins) float32     preprocess.py:13-18
    multichannel_stft(sig)            (samples, ch) -> (ch, T, bins) complex64     preprocess.py:21-36
    multichannel_complex_to_log_mel   (..., bins) complex -> (..., mel_bins) float32   preprocess.py:39-45
    calculate_scalar_of_tensor        per-mel mean / std           preprocess.py:48-57
    LogMelFrontEnd                    fused waveform -> (B, 1, T, mel) for the training loop

The module-level functions use `DEFAULT_CONFIG` (the reference's committed constants); build a
`LogMelFrontEnd(cfg)` for any other parameter set.  No CPU path: inputs are moved to the GPU.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ... import _lib as L
from ...engine import _stream
from .spectogram_configs import REF_NATIVE, SpectogramConfig

DEFAULT_CONFIG = REF_NATIVE


# ---- constants built once on the host (like the reference does at import time) -------------------
def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    lin = m * (200.0 / 3.0)
    log = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0))
    return np.where(m >= 15.0, log, lin)


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3.0)
    log = 15.0 + np.log(np.maximum(f, 1e-12) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def mel_filter_bank(cfg: SpectogramConfig) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) defaults (Slaney scale, slaney norm),
    transposed to (bins, mel_bins) float32 like MEL_FILTER_BANK_MATRIX."""
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(cfg.mel_min_freq), _slaney_hz_to_mel(cfg.fmax),
                                          cfg.mel_bins + 2))
    freqs = np.linspace(0.0, cfg.working_sample_rate / 2.0, cfg.bins)
    lo, ce, hi = edges[:-2, None], edges[1:-1, None], edges[2:, None]
    up = (freqs[None, :] - lo) / (ce - lo)
    down = (hi - freqs[None, :]) / (hi - ce)
    tri = np.clip(np.minimum(up, down), 0.0, None) * (2.0 / (hi - lo))
    return tri.astype(np.float32).T.copy()


def padded_window(cfg: SpectogramConfig) -> np.ndarray:
    w = np.zeros(cfg.NFFT, dtype=np.float64)
    left = (cfg.NFFT - cfg.frame_size) // 2
    w[left:left + cfg.frame_size] = np.hanning(cfg.frame_size)
    return w.astype(np.float32)


class LogMelFrontEnd:
    """GPU front-end for one parameter set.  wave (B, samples) float32 -> (B, 1, T, mel) float32."""

    def __init__(self, cfg: SpectogramConfig = DEFAULT_CONFIG, device="cuda", mean=None, std=None):
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the log-mel front-end runs on the MI355X only (device='cuda')")
        self.mel_np = mel_filter_bank(cfg)                                  # (bins, mel)
        melT = np.ascontiguousarray(self.mel_np.T)                          # (mel, bins)
        nz = melT > 0
        lo = np.where(nz.any(1), nz.argmax(1), 0).astype(np.int32)
        hi = np.where(nz.any(1), cfg.bins - nz[:, ::-1].argmax(1), 0).astype(np.int32)
        dev = self.device
        self.melT = torch.from_numpy(melT).to(dev)
        self.mel_lo = torch.from_numpy(lo).to(dev)
        self.mel_hi = torch.from_numpy(hi).to(dev)
        self.window = torch.from_numpy(padded_window(cfg)).to(dev)
        self.mean = None if mean is None else torch.as_tensor(mean, dtype=torch.float32).to(dev).contiguous()
        self.std = None if std is None else torch.as_tensor(std, dtype=torch.float32).to(dev).contiguous()
        self.ws = torch.empty(max(1, L.lib().sed_logmel_ws_bytes(1, 1, cfg.NFFT, cfg.hop_size) // 4),
                              dtype=torch.float32, device=dev)

    def _wave(self, wave) -> torch.Tensor:
        w = torch.as_tensor(wave)
        if w.dim() != 2:
            raise ValueError("expected (B, samples)")
        return w.to(self.device, dtype=torch.float32).contiguous()

    def __call__(self, wave, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        w = self._wave(wave)
        B, n = w.shape
        T = self.cfg.num_frames(n)
        if out is None:
            out = torch.empty((B, 1, T, self.cfg.mel_bins), dtype=torch.float32, device=self.device)
        L.check(L.lib().sed_logmel_fwd(L.ptr(w), L.ptr(self.window), L.ptr(self.melT), L.ptr(self.mel_lo),
                                       L.ptr(self.mel_hi), L.ptr(self.mean), L.ptr(self.std), L.ptr(out),
                                       L.ptr(self.ws), B, n, self.cfg.NFFT, self.cfg.hop_size, self.cfg.mel_bins,
                                       _stream()), "logmel_fwd")
        return out

    def stft(self, wave) -> torch.Tensor:
        w = self._wave(wave)
        B, n = w.shape
        T = self.cfg.num_frames(n)
        spec = torch.empty((B, T, self.cfg.bins), dtype=torch.complex64, device=self.device)
        L.check(L.lib().sed_stft_fwd(L.ptr(w), L.ptr(self.window), L.ptr(spec), L.ptr(self.ws), B, n, self.cfg.NFFT,
                                     self.cfg.hop_size, _stream()), "stft_fwd")
        return spec

    def complex_to_log_mel(self, spec: torch.Tensor, normalise: bool = False) -> torch.Tensor:
        s = torch.as_tensor(spec).to(self.device, dtype=torch.complex64).contiguous()
        if s.shape[-1] != self.cfg.bins:
            raise ValueError(f"last dim must be {self.cfg.bins} frequency bins")
        lead = s.shape[:-1]
        nframes = int(np.prod(lead)) if len(lead) else 1
        out = torch.empty(lead + (self.cfg.mel_bins,), dtype=torch.float32, device=self.device)
        L.check(L.lib().sed_complex_to_logmel(L.ptr(s), L.ptr(self.melT), L.ptr(self.mel_lo), L.ptr(self.mel_hi),
                                              L.ptr(self.mean) if normalise else None,
                                              L.ptr(self.std) if normalise else None, L.ptr(out), nframes,
                                              self.cfg.bins, self.cfg.mel_bins, _stream()), "complex_to_logmel")
        return out


class PrefetchingFrontEnd:
    """Software pipeline around a LogMelFrontEnd: the log-mel features of the NEXT batch are computed on a second
    HIP stream while the train step of the current batch runs (what the reference's DataLoader workers do on the
    host, train.py:94 / main.py:125).  The front-end is VALU-bound and uses neither the matrix pipe nor much HBM, so
    it fills issue slots the convolution kernels leave idle.

        pf.submit(wave)              # prime: features of batch 0
        for each step:
            x = pf.get()             # current stream waits for the oldest submitted batch
            pf.submit(next_wave)     # enqueue the following batch (waits until its buffer is released)
            trainer.train_step(x, y)
            pf.release()             # x may be overwritten from this point of the current stream on
    """

    def __init__(self, fe: "LogMelFrontEnd", nbuf: int = 2, stream=None):
        self.fe = fe
        # `stream`: e.g. a LOW-priority stream, so that the front-end's many short workgroups only fill the CUs the train step's
        # kernels leave free (its ~40 small launches per step) instead of competing with the 256-workgroup convolution kernels
        self.stream = stream if stream is not None else torch.cuda.Stream(device=fe.device)
        self.nbuf = int(nbuf)
        self.buf = [None] * self.nbuf
        self.done = [torch.cuda.Event() for _ in range(self.nbuf)]
        self.free = [None] * self.nbuf          # event of the consumer's release (None: never used)
        self.head = 0                           # next buffer to submit into
        self.tail = 0                           # next buffer to hand out
        self.pending = 0
        self.timer = None                       # optional engine.KernelTimer (events on the side stream)

    def submit(self, wave) -> None:
        if self.pending >= self.nbuf:
            raise RuntimeError("PrefetchingFrontEnd: every buffer is in flight (get/release one first)")
        k = self.head
        w = self.fe._wave(wave)
        T = self.fe.cfg.num_frames(w.shape[1])
        shape = (w.shape[0], 1, T, self.fe.cfg.mel_bins)
        if self.buf[k] is None or tuple(self.buf[k].shape) != shape:
            self.buf[k] = torch.empty(shape, dtype=torch.float32, device=self.fe.device)
        self.stream.wait_stream(torch.cuda.current_stream())      # the waveform was produced on the caller's stream
        if self.free[k] is not None:
            self.stream.wait_event(self.free[k])
        with torch.cuda.stream(self.stream):
            if self.timer is not None:
                self.timer.launch("sed_logmel_fwd", lambda: self.fe(w, out=self.buf[k]), ())
            else:
                self.fe(w, out=self.buf[k])
            self.done[k].record(self.stream)
        self.head = (k + 1) % self.nbuf
        self.pending += 1

    def get(self) -> torch.Tensor:
        if self.pending == 0:
            raise RuntimeError("PrefetchingFrontEnd: nothing submitted")
        k = self.tail
        torch.cuda.current_stream().wait_event(self.done[k])
        self._held = k
        self.tail = (k + 1) % self.nbuf
        self.pending -= 1
        return self.buf[k]

    def release(self) -> None:
        k = self._held
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.free[k] = ev


# ---- module-level API with the reference's names -------------------------------------------------
MEL_FILTER_BANK_MATRIX = mel_filter_bank(DEFAULT_CONFIG)
_default_fe: Optional[LogMelFrontEnd] = None


def _fe() -> LogMelFrontEnd:
    global _default_fe
    if _default_fe is None:
        _default_fe = LogMelFrontEnd(DEFAULT_CONFIG)
    return _default_fe


def multichannel_stft(multichannel_signal):
    """(samples, channels) array -> (channels, T, NFFT/2+1) complex64 numpy array."""
    sig = np.asarray(multichannel_signal)
    return _fe().stft(np.ascontiguousarray(sig.T)).cpu().numpy()


def multichannel_complex_to_log_mel(multichannel_complex_spectogram):
    """(..., NFFT/2+1) complex -> (..., mel_bins) float32 numpy array."""
    return _fe().complex_to_log_mel(np.asarray(multichannel_complex_spectogram)).cpu().numpy()


def calculate_scalar_of_tensor(x):
    """Per-mel mean / population std over (channels, frames): dataset statistics, computed once at
    preprocessing time on the host (preprocess.py:48-57)."""
    x = np.asarray(x)
    axis = 0 if x.ndim == 2 else (0, 1)
    return np.mean(x, axis=axis), np.std(x, axis=axis)


def preprocess_data(audio_path_and_labels, output_dir, output_mean_std_file, preprocess_mode="logMel",
                    cfg: SpectogramConfig = DEFAULT_CONFIG):
    """preprocess.py:60-88 minus the debug plot: per recording STFT (+ log-mel) on the MI355X, the
    reference's pickle layout on disk, then the dataset-wide mean / std."""
    import os
    import pickle
    from ..dataset_utils import read_multichannel_audio
    fe = LogMelFrontEnd(cfg)
    os.makedirs(output_dir, exist_ok=True)
    all_features = []
    for (audio_path, start_times, end_times, audio_name) in audio_path_and_labels:
        wave = read_multichannel_audio(audio_path=audio_path, target_fs=cfg.working_sample_rate, cfg=cfg)
        feature = fe.stft(np.ascontiguousarray(wave.T))                    # (ch, T, bins) complex64, device
        if preprocess_mode == "logMel":
            feature = fe.complex_to_log_mel(feature)
        feature = feature.cpu().numpy()
        all_features.append(feature)
        with open(os.path.join(output_dir, audio_name + f"_{preprocess_mode}_features_and_labels.pkl"), "wb") as f:
            pickle.dump({"features": feature, "start_times": start_times, "end_times": end_times}, f)
    mean, std = calculate_scalar_of_tensor(np.concatenate(all_features, axis=1))
    with open(output_mean_std_file, "wb") as f:
        pickle.dump({"mean": mean, "std": std}, f)
    return mean, std
