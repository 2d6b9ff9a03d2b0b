"""SpectogramDataset with the feature bank resident in HBM.

Counterpart of /root/reference/dataset/spectogram/spectograms_dataset.py:17-283.  Same constructor,
same on-disk format (the pickles `preprocess_data` writes: {'features', 'start_times',
'end_times'} per recording + {'mean', 'std'}), same protocol towards train()/eval():

    __len__, __getitem__(idx) -> (features (1, crop, mel) float32, event_matrix (crop, classes))
    get_validation_sampler(max_validate_num) -> (features (1, 1, T, mel), events (1, T, classes), name)

What differs is WHERE the per-sample work runs.  The reference concatenates all training features
in host memory and lets DataLoader workers crop / mix / add noise / z-score / (in 'Complex' mode)
convert to log-mel per sample on the CPU (:58-78, :104-135).  Here the concatenated bank is
uploaded once (a 100-recording REF-NATIVE complex bank is 2.4 GB of the 288 GB) and one kernel
launch produces a whole batch: `sed_logmel_crops` ('logMel') or `sed_complex_augment_logmel`
('Complex': gather + mix + noise + complex z-score + |.|^2 + mel + log).  `DeviceBatchLoader`
is the loader train() is meant to be fed with; `__getitem__` runs the same kernels with B = 1 and
returns CUDA tensors (use num_workers=0 with a torch DataLoader).

Host-side randomness follows the reference call for call (np.random.choice / randint / rand), so
the augmentation *decisions* are reproducible with np.random.seed; the Gaussian noise itself is
generated on the device from (seed, element index).
"""
from __future__ import annotations

import ctypes
import os
import pickle
from random import shuffle

import numpy as np
import torch
from torch.utils.data import Dataset

from ... import _lib as L
from ...engine import _stream
from .preprocess import LogMelFrontEnd
from .spectogram_configs import REF_NATIVE, SpectogramConfig


def create_event_matrix(frames_num, start_times, end_times, cfg: SpectogramConfig = REF_NATIVE):
    """(:202-215) per-frame 0/1 matrix, float64 like np.zeros' default."""
    event_matrix = np.zeros((frames_num, cfg.classes_num))
    for n in range(len(start_times)):
        start_frame = int(round(start_times[n] * cfg.frames_per_second))
        end_frame = int(round(end_times[n] * cfg.frames_per_second)) + 1
        event_matrix[start_frame:end_frame] = 1
    return event_matrix


def split_train_val(feature_names, val_descriptor):
    """(:268-283) float -> shuffled percentage split; str -> files containing the substring validate."""
    if type(val_descriptor) == float:
        shuffle(feature_names)
        val_split = int(len(feature_names) * val_descriptor)
        return feature_names[val_split:], feature_names[:val_split]
    train, val = [], []
    for name in feature_names:
        (val if val_descriptor in name else train).append(name)
    return train, val


def _load(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def _read_train_data_to_memory(train_feature_paths, crop_size, balance_classes, cfg):
    """(:138-185) concatenate all recordings on the frame axis and build the shuffled list of crop
    start indices (split by 'crop sees an event', optionally balanced)."""
    frame_index = 0
    feats, events, with_event, empty = [], [], [], []
    for path in train_feature_paths:
        data = _load(path)
        feature = data["features"]
        event_matrix = create_event_matrix(feature.shape[1], data["start_times"], data["end_times"], cfg)
        frames_num = feature.shape[1]
        possible = np.arange(frame_index, frame_index + frames_num - crop_size)
        frame_index += frames_num
        feats.append(feature)
        events.append(event_matrix)
        flag = np.zeros(possible.shape, dtype=bool)
        for i in np.where(event_matrix > 0)[0]:
            flag[i - crop_size: i] = True            # the reference's window, kept as is (:169-170)
        with_event += possible[np.where(flag)[0]].tolist()
        empty += possible[np.where(~flag)[0]].tolist()
    train_features = np.concatenate(feats, axis=1)
    train_event_matrix = np.concatenate(events, axis=0)
    np.random.shuffle(with_event)
    np.random.shuffle(empty)
    if balance_classes:
        size = min(len(with_event), len(empty))
        with_event, empty = with_event[:size], empty[:size]
    starts = np.concatenate((empty, with_event)).astype(np.int64)
    np.random.shuffle(starts)
    return train_features, train_event_matrix, starts


class SpectogramDataset(Dataset):
    def __init__(self, features_and_labels_dir, mean_std_file, val_descriptor, balance_classes=False,
                 augment_data=False, preprocessed_mode="Complex", cfg: SpectogramConfig = REF_NATIVE,
                 device="cuda", noise_seed=0):
        assert preprocessed_mode in ["logMel", "Complex"], "Spectogram type should be either logmel or complex"
        assert not (preprocessed_mode == "logMel" and augment_data), "Can't perform augmentation in logMel spectograms"
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SpectogramDataset keeps its feature bank on the MI355X (device='cuda'); "
                               "there is no CPU path")
        self.cfg = cfg
        self.preprocessed_mode = preprocessed_mode
        self.augment_data = augment_data
        self.train_crop_size = cfg.train_crop_size
        self.noise_seed = int(noise_seed)
        self._noise_calls = 0

        d = _load(mean_std_file)
        self.mean, self.std = d["mean"], d["std"]

        all_paths = [os.path.join(features_and_labels_dir, x) for x in sorted(os.listdir(features_and_labels_dir))]     # (sorted: the same order on every rank / file system)
        train_paths, self.val_feature_paths = split_train_val(all_paths, val_descriptor)
        feats, self.train_event_matrix, self.train_start_indices = _read_train_data_to_memory(
            train_paths, cfg.train_crop_size, balance_classes, cfg)
        if feats.shape[0] != 1:
            raise ValueError(f"the model consumes one audio channel (cfg.audio_channels == 1); got {feats.shape[0]}")
        self.val_features_list, self.val_event_matrix_list = [], []
        for p in self.val_feature_paths:
            data = _load(p)
            self.val_features_list.append(data["features"])
            self.val_event_matrix_list.append(create_event_matrix(data["features"].shape[1], data["start_times"],
                                                                  data["end_times"], cfg))

        dev = self.device
        self.fe = LogMelFrontEnd(cfg, device=dev)
        self.bank_frames = int(feats.shape[1])
        if preprocessed_mode == "logMel":
            self.bank = torch.from_numpy(np.ascontiguousarray(feats[0], dtype=np.float32)).to(dev)
            self.d_mean = torch.as_tensor(np.broadcast_to(self.mean, (cfg.mel_bins,)).astype(np.float32)).to(dev)
            self.d_std = torch.as_tensor(np.broadcast_to(self.std, (cfg.mel_bins,)).astype(np.float32)).to(dev)
        else:
            self.bank = torch.from_numpy(np.ascontiguousarray(feats[0], dtype=np.complex64)).to(dev)
            self.d_mean = torch.as_tensor(np.broadcast_to(self.mean, (cfg.bins,)).astype(np.complex64)).to(dev)
            self.d_std = torch.as_tensor(np.broadcast_to(self.std, (cfg.bins,)).astype(np.float32)).to(dev)
        self.d_events = torch.from_numpy(self.train_event_matrix).to(dev)          # float64, like the reference
        print(f"Data generator initiated with {len(train_paths)} train samples "
              f"totaling {len(self.train_event_matrix) / cfg.frames_per_second:.1f} seconds "
              f"and {len(self.val_feature_paths)} val samples; feature bank "
              f"{self.bank.numel() * self.bank.element_size() / 2**20:.1f} MiB on {dev}")

    def __len__(self):
        return len(self.train_start_indices)

    # ---- host-side augmentation decisions, drawn in the reference's order (:71-76, :112-135) -----------------
    def _draw(self, idx):
        starts = [int(self.train_start_indices[idx])]
        noise_std = 0.0
        if self.augment_data:
            n_aug = int(np.random.choice([0, 1, 2, 3], 1, p=[0.6, 0.25, 0.1, 0.05])[0])
            for _ in range(n_aug):
                # the reference draws randint(len + 1) and can index one past the end (:126); draw inside the table
                starts.append(int(self.train_start_indices[np.random.randint(len(self.train_start_indices))]))
            r = np.random.rand()
            if r > 0.5:
                noise_std = 0.001 + (r + 0.5) * (0.005 - 0.001)
        return starts, noise_std

    def device_batch(self, indices):
        """One launch for a whole batch: returns (features (B,1,crop,mel) float32, events (B,crop,classes)
        float64), both on the device."""
        B, crop, cfg = len(indices), self.train_crop_size, self.cfg
        draws = [self._draw(int(i)) for i in indices]
        starts_h = np.zeros((B, 4), dtype=np.int32)
        nmix_h = np.ones(B, dtype=np.int32)
        nstd_h = np.zeros(B, dtype=np.float32)
        for b, (st, ns) in enumerate(draws):
            starts_h[b, :len(st)] = st
            nmix_h[b] = len(st)
            nstd_h[b] = ns
        dev = self.device
        starts_d = torch.from_numpy(starts_h).to(dev)
        out = torch.empty((B, 1, crop, cfg.mel_bins), dtype=torch.float32, device=dev)
        if self.preprocessed_mode == "logMel":
            s1 = np.ascontiguousarray(starts_h[:, 0])
            s1_d = starts_d[:, 0].contiguous()
            L.check(L.lib().sed_logmel_crops(L.ptr(self.bank), self.bank_frames, s1.ctypes.data, L.ptr(s1_d),
                                             L.ptr(self.d_mean), L.ptr(self.d_std), L.ptr(out), B, crop, cfg.mel_bins,
                                             _stream()), "logmel_crops")
        else:
            nmix_d = torch.from_numpy(nmix_h).to(dev)
            nstd_d = torch.from_numpy(nstd_h).to(dev)
            self._noise_calls += 1
            seed = (self.noise_seed * 0x9E3779B1 + self._noise_calls) & 0xFFFFFFFFFFFFFFFF
            fe = self.fe
            L.check(L.lib().sed_complex_augment_logmel(
                L.ptr(self.bank), self.bank_frames, starts_h.ctypes.data, nmix_h.ctypes.data, L.ptr(starts_d),
                L.ptr(nmix_d), L.ptr(nstd_d), None, ctypes.c_ulonglong(seed), L.ptr(self.d_mean), L.ptr(self.d_std),
                L.ptr(fe.melT), L.ptr(fe.mel_lo), L.ptr(fe.mel_hi), L.ptr(out), B, crop, cfg.bins, cfg.mel_bins,
                _stream()), "complex_augment_logmel")
        # labels: gather + max over the mixed crops (:132) -- index plumbing on the label bank
        ar = torch.arange(crop, device=dev)
        st = starts_d.long()
        ev = self.d_events[st[:, 0:1] + ar[None, :]]
        for j in range(1, 4):
            sel = torch.from_numpy(nmix_h > j).to(dev)
            if bool((nmix_h > j).any()):
                evj = self.d_events[st[:, j:j + 1] + ar[None, :]]
                ev = torch.where(sel[:, None, None], torch.maximum(ev, evj), ev)
        return out, ev

    def __getitem__(self, idx):
        f, e = self.device_batch([idx])
        return f[0], e[0]

    def transform(self, x):
        """(:104-110) for a whole recording: (channels, frames, mel | bins) array -> (channels, frames, mel) CUDA."""
        x = np.asarray(x)
        ch, T = x.shape[0], x.shape[1]
        dev, cfg = self.device, self.cfg
        out = torch.empty((ch, T, cfg.mel_bins), dtype=torch.float32, device=dev)
        for c in range(ch):
            starts_h = np.zeros((1, 4), dtype=np.int32)
            starts_d = torch.zeros((1, 4), dtype=torch.int32, device=dev)
            if self.preprocessed_mode == "logMel":
                bank = torch.from_numpy(np.ascontiguousarray(x[c], dtype=np.float32)).to(dev)
                L.check(L.lib().sed_logmel_crops(L.ptr(bank), T, starts_h.ctypes.data, L.ptr(starts_d), L.ptr(self.d_mean),
                                                 L.ptr(self.d_std), L.ptr(out[c]), 1, T, cfg.mel_bins, _stream()),
                        "logmel_crops")
            else:
                bank = torch.from_numpy(np.ascontiguousarray(x[c], dtype=np.complex64)).to(dev)
                one_h = np.ones(1, dtype=np.int32)
                one_d = torch.ones(1, dtype=torch.int32, device=dev)
                fe = self.fe
                L.check(L.lib().sed_complex_augment_logmel(
                    L.ptr(bank), T, starts_h.ctypes.data, one_h.ctypes.data, L.ptr(starts_d), L.ptr(one_d), None, None,
                    ctypes.c_ulonglong(0), L.ptr(self.d_mean), L.ptr(self.d_std), L.ptr(fe.melT), L.ptr(fe.mel_lo),
                    L.ptr(fe.mel_hi), L.ptr(out[c]), 1, T, cfg.bins, cfg.mel_bins, _stream()), "complex_augment_logmel")
        return out

    def get_validation_sampler(self, max_validate_num=None):
        """(:80-102) whole recordings, batch 1."""
        for n in range(len(self.val_feature_paths)):
            if n == max_validate_num:
                break
            name = os.path.basename(os.path.splitext(self.val_feature_paths[n])[0])
            feature = self.transform(self.val_features_list[n])
            yield feature[None], torch.from_numpy(self.val_event_matrix_list[n][None]), name


class DeviceBatchLoader:
    """What train() iterates over instead of torch's DataLoader: walks the pre-shuffled start-index
    table in order (the reference builds DataLoader without shuffle, main.py:125), `batch_size`
    crops per launch; with world_size > 1 rank r takes the slice
    idx = step*B_global + r*B_local + i  (SURVEY 8e).  One process keeps the last, short batch like DataLoader; with
    world_size > 1 the ragged tail of the last global batch wraps to the start of the table so that every rank holds a full
    batch (train.sharded_batch_indices: no crop is dropped from an epoch, len() = ceil(n / B_global) at every world size)."""

    def __init__(self, dataset, batch_size, rank=0, world_size=1):
        self.dataset, self.batch_size, self.rank, self.world_size = dataset, int(batch_size), int(rank), int(world_size)

    def __len__(self):
        g = self.batch_size * self.world_size
        return (len(self.dataset) + g - 1) // g

    def __iter__(self):
        from ...train import sharded_batch_indices
        for idx in sharded_batch_indices(len(self.dataset), self.batch_size, self.rank, self.world_size):
            yield self.dataset.device_batch(idx)


def _processed_dirs(root, descriptor, mode, suffix=""):
    base = os.path.join(root, "processed", descriptor)
    return f"{base}/{mode}-features_and_labels{suffix}", f"{base}/{mode}-features_mean_std{suffix}.pkl"


def preprocess_tau_sed_data(data_dir, preprocess_mode, force_preprocess=False, fold_name="eval",
                            cfg: SpectogramConfig = REF_NATIVE, labels=("doorslam",)):
    """(:218-239) without the download step (no network on either box): expects the extracted
    TAU-SED-2019 tree under data_dir/Tau_sound_events_2019 and (re)builds the feature pickles."""
    from ..dataset_utils import get_tau_sed_paths_and_labels, tau_audio_and_meta_dirs
    from .preprocess import preprocess_data
    root = f"{data_dir}/Tau_sound_events_2019"
    descriptor = cfg.cfg_descriptor + f"_C-{'-'.join(labels)}"
    feats_dir, mean_std = _processed_dirs(root, descriptor, preprocess_mode, f"_{fold_name}")
    if not os.path.exists(feats_dir) or force_preprocess:
        audio_dir, meta_dir = tau_audio_and_meta_dirs(root, fold_name)
        preprocess_data(get_tau_sed_paths_and_labels(audio_dir, meta_dir, labels), output_dir=feats_dir,
                        output_mean_std_file=mean_std, preprocess_mode=preprocess_mode, cfg=cfg)
    else:
        print("Using existing mel features")
    return feats_dir, mean_std


def preprocess_film_clap_data(data_dir, preprocessed_mode, force_preprocess=False, cfg: SpectogramConfig = REF_NATIVE,
                              time_margin=0.33):
    """(:242-265)"""
    from ..dataset_utils import get_film_clap_paths_and_labels
    from .preprocess import preprocess_data
    film_clap_dir = os.path.join(data_dir, "FilmClap")
    if not os.path.exists(film_clap_dir):
        raise Exception("You should get you own dataset...")
    descriptor = cfg.cfg_descriptor + f"_tm-{time_margin}"
    feats_dir, mean_std = _processed_dirs(film_clap_dir, descriptor, preprocessed_mode)
    if not os.path.exists(feats_dir) or force_preprocess:
        print("preprocessing raw data")
        preprocess_data(get_film_clap_paths_and_labels(film_clap_dir, time_margin=time_margin), output_dir=feats_dir,
                        output_mean_std_file=mean_std, preprocess_mode=preprocessed_mode, cfg=cfg)
    else:
        print("Using existing mel features")
    return feats_dir, mean_std
