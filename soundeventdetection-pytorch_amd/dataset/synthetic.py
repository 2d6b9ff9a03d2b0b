"""A synthetic, in-memory dataset with the SpectogramDataset protocol of
/root/reference/dataset/spectogram/spectograms_dataset.py:17-102 (what train()/eval() consume):

    __len__, __getitem__(idx) -> (features (1, crop, mel) float32, event_matrix (crop, classes) float64)
    get_validation_sampler(max_validate_num) -> yields (features (1, 1, T, mel), events (1, T, classes), name)

TAU-SED-2019 cannot be downloaded on either box (no network), so parity of the *training outcome*
(frame-F1) is demonstrated on this seeded task: z-scored noise "log-mel" frames with a
band-limited energy bump wherever an event is active.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset


def _events(rng, T, classes, rate=0.04, min_run=10):
    y = np.zeros((T, classes), dtype=np.float64)
    n_runs = max(1, int(round(rate * T / (1.5 * min_run))))
    for k in range(classes):
        for _ in range(n_runs):
            s = int(rng.integers(0, max(1, T - 2 * min_run)))
            y[s:s + min_run + int(rng.integers(0, min_run)), k] = 1.0
    return y


def _features(rng, y, mel_bins, snr=1.6):
    T, K = y.shape
    x = rng.standard_normal((T, mel_bins)).astype(np.float32)
    for k in range(K):
        lo = (8 + 11 * k) % (mel_bins - 16)
        band = np.hanning(16).astype(np.float32)
        x[:, lo:lo + 16] += snr * y[:, k:k + 1].astype(np.float32) * band[None, :]
    return x


class SyntheticSedDataset(Dataset):
    def __init__(self, n_train_crops=256, crop=240, n_val=6, val_frames=808, mel_bins=64, classes=1, seed=0):
        rng = np.random.default_rng(seed)
        self.crop, self.mel_bins, self.classes = crop, mel_bins, classes
        self.train = []
        for _ in range(n_train_crops):
            y = _events(rng, crop, classes, rate=0.15, min_run=8)
            self.train.append((_features(rng, y, mel_bins)[None], y))
        self.val = []
        for i in range(n_val):
            y = _events(rng, val_frames, classes, rate=0.15, min_run=8)
            self.val.append((_features(rng, y, mel_bins)[None], y, f"synthetic_val_{i}"))

    def __len__(self):
        return len(self.train)

    def __getitem__(self, idx):
        f, y = self.train[idx]
        return torch.from_numpy(f), torch.from_numpy(y)

    def get_validation_sampler(self, max_validate_num=None):
        for n, (f, y, name) in enumerate(self.val):
            if n == max_validate_num:
                break
            yield torch.from_numpy(f[None]), torch.from_numpy(y[None]), name
