import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# A/B switches of the library are flipped through its C ABI (qtts_set_option, include/qtts.h) -- the `qopt` fixture below -- never through
# os.environ inside the process: the library looks at the environment once per switch.


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


@pytest.fixture
def qopt():
    """qopt(lib, "QTTS_X", value): set an A/B switch of `lib` (a ctypes handle of libqtts / the host-emulation build) through the C ABI;
    value None removes the override.  Every switch set through the fixture is cleared again when the test ends."""
    import ctypes as C
    touched = []

    def set_(lib, name, value):
        lib.qtts_set_option.argtypes = [C.c_char_p, C.c_char_p]
        lib.qtts_set_option.restype = C.c_int
        rc = lib.qtts_set_option(name.encode(), None if value is None else str(value).encode())
        assert rc == 0, (name, value, rc)
        touched.append((lib, name))

    yield set_
    for lib, name in touched:
        lib.qtts_set_option(name.encode(), None)
