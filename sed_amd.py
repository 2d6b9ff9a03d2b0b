"""Importable alias of the hyphenated package directory `soundeventdetection-pytorch_amd/`."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("soundeventdetection-pytorch_amd")
