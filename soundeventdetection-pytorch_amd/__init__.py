"""MI355X-native drop-in for the spectrogram training path of ariel415el/SoundEventDetection-Pytorch.

The directory name follows the build contract (`soundeventdetection-pytorch_amd`); because of the
hyphen it is imported with importlib (or through the `sed_amd` alias module at the repo root):

    import importlib; sed = importlib.import_module("soundeventdetection-pytorch_amd")
"""
from . import _lib  # noqa: F401
from .engine import CnnEngine  # noqa: F401
from .models.spectogram_models import (Cnn_AvgPooling, ConvBlock, Crnn_AvgPooling, init_bn,  # noqa: F401
                                       init_layer, interpolate)
from .models.waveform_models import M5  # noqa: F401
from .m5_engine import M5Engine  # noqa: F401
from .utils.common import WeightedBCE  # noqa: F401
from . import train  # noqa: F401,E402
from .train import FusedTrainer, FusedAdamAmsgrad  # noqa: F401,E402

__all__ = ["M5", "Cnn_AvgPooling", "Crnn_AvgPooling", "ConvBlock", "WeightedBCE", "CnnEngine", "interpolate", "init_layer", "init_bn"]
