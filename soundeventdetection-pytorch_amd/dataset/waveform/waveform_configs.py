"""Counterpart of /root/reference/dataset/waveform/waveform_configs.py + dataset/common_config.py:1-7."""
time_margin = 0.33
working_sample_rate = 48000
frame_size = int(working_sample_rate * time_margin * 2)
hop_size = frame_size // 2
audio_channels = 1
min_event_percentage_in_positive_frame = 0.74
frames_per_second = working_sample_rate // hop_size
classes_num = 1


def _human(n):
    for unit, div in (("M", 1e6), ("K", 1e3)):
        if abs(n) >= div:
            return f"{n / div:.1f}{unit}"
    return str(n)


cfg_descriptor = (f"WaveForm_SaR-{_human(working_sample_rate)}_FrS-{_human(frame_size)}"
                  f"_HoS-{_human(hop_size)}_Ch-{audio_channels}")
