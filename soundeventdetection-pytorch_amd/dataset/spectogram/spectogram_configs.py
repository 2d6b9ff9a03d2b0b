"""Runtime counterpart of /root/reference/dataset/common_config.py:2-16 and
/root/reference/dataset/spectogram/spectogram_configs.py:5-14 (module-level constants there, an
explicit object here: SURVEY D3 needs both the committed 48 kHz constants and the 32 kHz bench set)."""
from __future__ import annotations

import math
from dataclasses import dataclass


@dataclass(frozen=True)
class SpectogramConfig:
    working_sample_rate: int
    frame_size: int
    hop_size: int
    NFFT: int
    mel_bins: int = 64
    mel_min_freq: float = 20.0
    mel_max_freq: float = None        # default working_sample_rate // 2
    audio_channels: int = 1
    classes_num: int = 1              # len(tau_sed_labels) == 1 ('doorslam'), common_config.py:15-16

    @property
    def fmax(self) -> float:
        return float(self.working_sample_rate // 2) if self.mel_max_freq is None else float(self.mel_max_freq)

    @property
    def bins(self) -> int:
        return self.NFFT // 2 + 1

    @property
    def frames_per_second(self) -> int:
        return self.working_sample_rate // self.hop_size

    @property
    def train_crop_size(self) -> int:
        return self.frames_per_second * 10

    def num_frames(self, samples: int) -> int:
        return 1 + samples // self.hop_size

    @property
    def cfg_descriptor(self) -> str:
        return (f"Spectogram_SaR-{self.working_sample_rate}_FrS-{self.frame_size}_HoS-{self.hop_size}"
                f"_Mel-{self.mel_bins}_Ch-{self.audio_channels}")


def reference_config(time_margin: float = 0.33, working_sample_rate: int = 48000) -> SpectogramConfig:
    frame_size = int(working_sample_rate * time_margin * 2)
    nfft = 2 ** int(math.ceil(math.log2(frame_size)))
    return SpectogramConfig(working_sample_rate, frame_size, frame_size // 2, nfft)


def bench_config() -> SpectogramConfig:
    """32 kHz, window = NFFT = 1024, hop 320: 6001 frames per 60 s clip (BASELINE.json configs[1])."""
    return SpectogramConfig(32000, 1024, 320, 1024)


REF_NATIVE = reference_config()
BENCH = bench_config()
