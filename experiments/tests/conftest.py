import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in (ROOT, os.path.join(ROOT, "tests")):
    if d not in sys.path:
        sys.path.insert(0, d)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X")
    config.addinivalue_line("markers", "experiments: needs libsed_hip.so built with `make EXPERIMENTS=1`")


@pytest.fixture(autouse=True)
def _sed_env_cache():
    """libsed_hip.so caches its SED_* knobs per name: drop the cache after every test (as tests/conftest.py does)."""
    yield
    mod = sys.modules.get("soundeventdetection-pytorch_amd._lib")
    if mod is not None and getattr(mod, "_lib", None) is not None:
        mod._lib.sed_config_reload()
