"""Counterpart of /root/reference/dataset/waveform/waveform_dataset.py (split_to_frames_with_hop_size :9-31,
get_start_indices_labesl :34-44, WaveformDataset :47-139, split_train_val :142-159).

Same protocol for train()/eval(): `__len__`, `__getitem__ -> (waveform (1, frame_size), label)`,
`get_validation_sampler(max_validate_num) -> (frames (n, 1, frame_size), labels (n,), name)`.  MI355X-first
difference: the concatenated training waveform and the per-start-index labels also live on the GPU, and
`device_batch(indices)` gathers (and augments) a whole batch of frames there -- WaveformBatchLoader feeds the
trainer without a host round trip.  Audio can come from files (dataset_utils.read_multichannel_audio) or from
in-memory arrays (`waveforms=`), which is what the synthetic task uses."""
from __future__ import annotations

import numpy as np
import torch

from . import waveform_configs as cfg


def split_to_frames_with_hop_size(waveform, start_times, end_times):
    """(:9-31) overlapping frames + a label per frame: covered by one event for more than
    min_event_percentage_in_positive_frame of its length."""
    frames, labels = [], []
    half = cfg.frame_size // 2
    for center in np.arange(half, waveform.shape[1] - half + 1, step=cfg.hop_size):
        frame = waveform[:, center - half: center + half]
        label = False
        for s, e in zip(start_times, end_times):
            lo = max(s * cfg.working_sample_rate, center - half)
            hi = min(e * cfg.working_sample_rate, center + half)
            label = label or (hi - lo) / cfg.frame_size > cfg.min_event_percentage_in_positive_frame
        frames.append(frame)
        labels.append(label)
    return frames, labels


def get_start_indices_labesl(waveform_length, start_times, end_times):
    """(:34-44) label[i] = 1 when a frame STARTING at sample i is covered enough by an event."""
    label = np.zeros(waveform_length)
    for start, end in zip(start_times, end_times):
        first = int(start * cfg.working_sample_rate - cfg.frame_size * (1 - cfg.min_event_percentage_in_positive_frame))
        last = int(end * cfg.working_sample_rate - cfg.frame_size * cfg.min_event_percentage_in_positive_frame)
        label[max(first, 0): max(last, 0)] = 1
    return label


def split_train_val(tuples, val_descriptor):
    """(:142-159)"""
    if type(val_descriptor) == float:
        np.random.shuffle(tuples)
        k = int(len(tuples) * val_descriptor)
        return tuples[k:], tuples[:k]
    train, val = [], []
    for t in tuples:
        (val if val_descriptor in t[0] else train).append(t)
    return train, val


class WaveformDataset:
    def __init__(self, audio_paths_labels_and_names, val_descriptor=0.15, balance_classes=False, augment_data=False,
                 waveforms=None, device=None):
        """`waveforms`: optional {audio_path_or_key: (channels, samples) array} replacing file reads."""
        self.balance_classes, self.augment_data = balance_classes, augment_data
        self.device = torch.device(device) if device is not None else None
        print("WaveformDataset:")
        train_items, val_items = split_train_val(list(audio_paths_labels_and_names), val_descriptor)

        def read(path):
            if waveforms is not None:
                return np.asarray(waveforms[path], dtype=np.float64)
            from ..dataset_utils import read_multichannel_audio
            return read_multichannel_audio(path, target_fs=cfg.working_sample_rate).T

        long_waveform, labels, starts = [], [], []
        frame_index = 0
        for (path, start_times, end_times, name) in train_items:
            w = read(path)
            long_waveform.append(w)
            starts.append(np.arange(frame_index, frame_index + w.shape[1] - cfg.frame_size, dtype=np.uint32))
            frame_index += w.shape[1]
            labels.append(get_start_indices_labesl(w.shape[1], start_times, end_times).astype(bool))
        self.long_waveform = np.concatenate(long_waveform, axis=1)
        self.all_start_indices_labels = np.concatenate(labels)
        self.possible_start_indices = np.concatenate(starts)
        np.random.shuffle(self.possible_start_indices)
        self.val_samples_sets, self.val_label_sets, self.val_file_names = [], [], []
        for (path, start_times, end_times, name) in val_items:
            frames, lab = split_to_frames_with_hop_size(read(path), start_times, end_times)
            self.val_samples_sets.append(frames)
            self.val_label_sets.append(lab)
            self.val_file_names.append(name)
        n = max(1, len(self.possible_start_indices))
        print(f"\t- Train split: {len(self.possible_start_indices)} overlapping fames. "
              f"~{100 * np.sum(self.all_start_indices_labels == 1) / n:.1f}% tagged as event")
        print(f"\t- Val split: {np.sum([len(x) for x in self.val_label_sets])} frames. "
              f"{np.sum([np.sum(x) for x in self.val_label_sets])} tagged as event")
        self._dev_wave = self._dev_labels = self._dev_starts = None

    # ---- reference protocol (host) ---------------------------------------------------------------------
    def get_validation_sampler(self, max_validate_num):
        for i, (frames, labels, name) in enumerate(zip(self.val_samples_sets, self.val_label_sets, self.val_file_names)):
            if max_validate_num is not None and i > max_validate_num:
                break
            yield torch.tensor(np.array(frames)), torch.tensor(np.array(labels)), name

    def __len__(self):
        return len(self.possible_start_indices)

    def __getitem__(self, idx):
        start = int(self.possible_start_indices[idx])
        waveform = self.long_waveform[:, start + np.arange(cfg.frame_size)].copy()
        label = self.all_start_indices_labels[start]
        if self.augment_data:
            waveform, label = self.augment_mix_samples(waveform, label)
            waveform, label = self.augment_add_noise(waveform, label)
        return waveform, label

    def augment_mix_samples(self, waveform, label):
        """(:122-129)"""
        k = np.random.choice([0, 1, 2, 3], 1, p=[0.5, 0.3, 0.15, 0.05])[0]
        for _ in range(k):
            j = int(np.random.choice(self.possible_start_indices))
            waveform += self.long_waveform[:, j + np.arange(cfg.frame_size)]
            label = max(label, self.all_start_indices_labels[j])
        waveform /= (k + 1)
        return waveform, label

    def augment_add_noise(self, waveform, label):
        """(:131-137)"""
        r = np.random.rand()
        if r > 0.5:
            noise_var = 0.001 + (r + 0.5) * (0.005 - 0.001)
            waveform += np.random.normal(0, noise_var, size=waveform.shape)
        return waveform, label

    # ---- device-side batches ---------------------------------------------------------------------------
    def _to_device(self, device):
        if self._dev_wave is None or self._dev_wave.device != device:
            self._dev_wave = torch.from_numpy(self.long_waveform.astype(np.float32)).to(device)
            self._dev_labels = torch.from_numpy(self.all_start_indices_labels.astype(np.float32)).to(device)
            self._dev_starts = torch.from_numpy(self.possible_start_indices.astype(np.int64)).to(device)

    def device_batch(self, indices: torch.Tensor, generator: torch.Generator = None):
        """(frames (B, 1, frame_size) float32, labels (B,) float32) on the GPU for dataset indices `indices`; the
        mix / noise augmentations of :122-137 run batched on the device (their random draws come from `generator`,
        not from numpy's global state)."""
        dev = indices.device
        self._to_device(dev)
        ar = torch.arange(cfg.frame_size, device=dev)
        starts = self._dev_starts[indices]
        x = self._dev_wave[:, starts[:, None] + ar[None, :]].permute(1, 0, 2).contiguous()      # (B, ch, frame)
        y = self._dev_labels[starts]
        if self.augment_data:
            B = indices.numel()
            k = torch.multinomial(torch.tensor([0.5, 0.3, 0.15, 0.05], device=dev), B, replacement=True, generator=generator)
            for i in range(1, 4):
                sel = (k >= i).nonzero().flatten()
                if sel.numel():
                    j = self._dev_starts[torch.randint(0, len(self), (sel.numel(),), device=dev, generator=generator)]
                    x[sel] += self._dev_wave[:, j[:, None] + ar[None, :]].permute(1, 0, 2)
                    y[sel] = torch.maximum(y[sel], self._dev_labels[j])
            x /= (k + 1).to(x.dtype)[:, None, None]
            r = torch.rand(B, device=dev, generator=generator)
            std = torch.where(r > 0.5, 0.001 + (r + 0.5) * (0.005 - 0.001), torch.zeros_like(r))
            x += torch.randn(x.shape, device=dev, generator=generator) * std[:, None, None]
        return x, y


class WaveformBatchLoader:
    """DataLoader stand-in: batches of frames gathered on the GPU; shards the (pre-shuffled) index sequence across
    data-parallel ranks as idx = step*B_global + rank*B_local + i.  Batches are multiples of 8 frames, so -- unlike the
    spectrogram loaders (train.sharded_batch_indices: ragged tail kept by one process, wrapped to the head of the table under data
    parallel) -- this loader DROPS the ragged tail of an epoch at every world size: len() = floor(n / B_global), the last
    n mod B_global frames of the (re-shuffled every epoch) index sequence are not visited in that epoch.  Train-only: validation
    walks whole files (WaveformDataset.get_validation_sampler)."""

    def __init__(self, dataset: WaveformDataset, batch_size: int, rank: int = 0, world_size: int = 1, device="cuda"):
        if batch_size % 8:
            raise ValueError("the M5 path needs batches that are a multiple of 8 frames")
        self.dataset, self.batch_size, self.rank, self.world = dataset, batch_size, rank, world_size
        self.device = torch.device(device)

    def __len__(self):
        return len(self.dataset) // (self.batch_size * self.world)

    def __iter__(self):
        B, n = self.batch_size, len(self)
        if n == 0:      # (the reference's DataLoader would yield one short batch; the M5 kernels need multiples of 8 per rank)
            raise ValueError(f"WaveformBatchLoader: {len(self.dataset)} training frames are fewer than one global batch "
                             f"({B} x {self.world} ranks): lower --batch_size")
        for step in range(n):
            base = step * B * self.world + self.rank * B
            yield self.dataset.device_batch(torch.arange(base, base + B, device=self.device))


def synthetic_waveform_task(n_files=6, seconds=20.0, seed=0):
    """A seeded stand-in for TAU / FilmClap (neither can be downloaded here): low-level noise with a 1.5 s
    tone burst per event.  Returns (audio_paths_labels_and_names, waveforms) for WaveformDataset(waveforms=...)."""
    rng = np.random.default_rng(seed)
    sr = cfg.working_sample_rate
    items, waves = [], {}
    for f in range(n_files):
        n = int(seconds * sr)
        w = rng.standard_normal(n) * 0.02
        starts, ends = [], []
        t = 1.0 + rng.random() * 2
        while t + 1.5 < seconds - 1:
            s, e = t, t + 1.5
            idx = np.arange(int(s * sr), int(e * sr))
            w[idx] += 0.3 * np.sin(2 * np.pi * (600 + 200 * rng.random()) * idx / sr) * np.hanning(idx.size)
            starts.append(s)
            ends.append(e)
            t = e + 1.5 + rng.random() * 3
        key = f"synthetic/{'val' if f == 0 else 'train'}_{f}.wav"
        items.append((key, np.array(starts), np.array(ends), f"synthetic_{f}"))
        waves[key] = w[None, :]
    return items, waves
