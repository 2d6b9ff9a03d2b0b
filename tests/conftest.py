import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(autouse=True)
def _sed_env_cache():
    """libsed_hip.so caches its SED_* knobs per name; monkeypatch restores the environment at teardown, so drop the
    cache after every test (only if the library is already loaded: CPU-only tests never load it)."""
    yield
    mod = sys.modules.get("soundeventdetection-pytorch_amd._lib")
    if mod is not None and getattr(mod, "_lib", None) is not None:
        mod._lib.sed_config_reload()
