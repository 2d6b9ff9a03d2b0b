"""Counterpart of /root/reference/dataset/dataset_utils.py:13-91 (collecting audio paths + labels,
reading audio).  Host-side file handling only -- no arithmetic of the training path lives here.

Audio decoding: the reference uses `soundfile` + `librosa.resample` (neither is in this image);
this module reads PCM/float WAV with scipy.io.wavfile and resamples with a polyphase filter
(scipy.signal.resample_poly).  The resampler is NOT librosa's (kaiser_best / soxr): features of
audio that needs resampling differ from the reference's at the filter-design level (documented
deviation; audio already at the working sample rate is bit-identical after decoding)."""
from __future__ import annotations

import json
import os
from collections import defaultdict
from fractions import Fraction

import numpy as np

from .spectogram.spectogram_configs import REF_NATIVE


def get_film_clap_paths_and_labels(data_root, time_margin=0.1):
    """(:13-41) [(audio_path, start_times, end_times, name)] from paths_and_labels_fixed_Meron.txt."""
    result, num_claps = [], 0
    files_per_film = defaultdict(int)
    path_to_label = json.load(open(os.path.join(data_root, "paths_and_labels_fixed_Meron.txt")))
    print("Collecting Film-clap dataset")
    for sound_path, centers in path_to_label.items():
        film_name = os.path.basename(os.path.dirname(sound_path))
        name = f"{film_name}_{os.path.splitext(os.path.basename(sound_path))[0]}"
        assert os.path.exists(sound_path), sound_path
        result.append((sound_path, [e - time_margin for e in centers], [e + time_margin for e in centers], name))
        num_claps += len(centers)
        files_per_film[film_name] += 1
    for film_name, n in files_per_film.items():
        print(f"\t- {film_name} has {n}")
    print(f"\tFilm clap dataset contains {len(result)} audio files with {num_claps} clap incidents")
    return result


def get_tau_sed_paths_and_labels(audio_dir, labels_data_dir, labels=("doorslam",)):
    """(:44-62) one csv per recording; keep the rows whose sound_event_recording is in `labels`."""
    import pandas as pd
    results = []
    for audio_fname in os.listdir(audio_dir):
        bare_name = os.path.splitext(audio_fname)[0]
        df = pd.read_csv(os.path.join(labels_data_dir, bare_name + ".csv"), sep=",")
        keep = [i for i, v in enumerate(df["sound_event_recording"].values) if v in labels]
        results.append((os.path.join(audio_dir, audio_fname), df["start_time"].values[keep],
                        df["end_time"].values[keep], bare_name))
    return results


def tau_audio_and_meta_dirs(root, fold_name="eval"):
    """Where download_tau_sed_2019.ensure_tau_data leaves the extracted data; no download here."""
    audio_dir = os.path.join(root, "raw", f"foa_{fold_name}")                 # download_tau_sed_2019.py:58-60
    meta_dir = os.path.join(root, "raw", f"metadata_{fold_name}")
    for d in (audio_dir, meta_dir):
        if not os.path.isdir(d):
            raise FileNotFoundError(f"{d} is missing: TAU-SED-2019 has to be placed there by hand "
                                    "(this build never downloads; there is no network on the GPU boxes)")
    return audio_dir, meta_dir


def _decode_wav(path):
    from scipy.io import wavfile
    sample_rate, data = wavfile.read(path)
    if data.dtype == np.uint8:
        audio = (data.astype(np.float64) - 128.0) / 128.0
    elif np.issubdtype(data.dtype, np.integer):
        audio = data.astype(np.float64) / float(2 ** (8 * data.dtype.itemsize - 1))   # soundfile's float64 scaling
    else:
        audio = data.astype(np.float64)
    return audio, int(sample_rate)


def read_multichannel_audio(audio_path, target_fs=None, cfg=REF_NATIVE):
    """(:65-91) (samples, channels) float64 at target_fs with cfg.audio_channels channels."""
    audio, sample_rate = _decode_wav(audio_path)
    if audio.ndim == 1:
        audio = audio.reshape(-1, 1)
    if audio.shape[1] < cfg.audio_channels:
        audio = np.repeat(audio.mean(1).reshape(-1, 1), cfg.audio_channels, axis=1)
    elif cfg.audio_channels == 1:
        audio = audio.mean(1).reshape(-1, 1)
    elif audio.shape[1] > cfg.audio_channels:
        audio = audio[:, :cfg.audio_channels]
    if target_fs is not None and sample_rate != target_fs:
        from scipy.signal import resample_poly
        fr = Fraction(int(target_fs), int(sample_rate))
        audio = np.stack([resample_poly(audio[:, i], fr.numerator, fr.denominator) for i in range(audio.shape[1])],
                         axis=1)
    return audio
