import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the library copies its A/B environment switches once (csrc/common.h QTTS_ENV); tests flip them with monkeypatch inside one process
os.environ.setdefault("QTTS_DEBUG_ENV_LIVE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def libqtts():
    """Path of the built C-ABI library (built on demand: hipcc cross-compiles without a GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("qtts_build", os.path.join(ROOT, "qwen3-tts_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.build(verbose=False)
