"""TEST INFRASTRUCTURE: lets the product's Python layer run against the host-emulation build of libqtts (tests/hostemu).

`install()` points the product's loader at libqtts_hostemu.so (QTTS_LIBRARY), tells the one place that insists on a HIP device
(`_lib.hip_device`) to hand out the CPU device, and replaces the handful of `torch.cuda.*` stream / context calls the engine
classes make by inert stand-ins.  `uninstall()` undoes all of it.  Used by tests/test_glue_on_emulator.py and by the
launcher test of bench.py (`QTTS_BENCH_HOSTEMU=1`, which marks its output line as not-a-measurement).  Nothing of this is
reachable from the product: the product refuses a CPU device.
"""
import contextlib
import os
import sys
from unittest import mock

HERE = os.path.dirname(os.path.abspath(__file__))
_STATE = None


class _FakeStream:
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def wait_stream(self, other):
        pass

    def synchronize(self):
        pass


class _NullDeviceCtx(contextlib.AbstractContextManager):
    def __init__(self, *a, **k):
        pass

    def __exit__(self, *exc):
        return False


def install():
    """Build (if stale) and select the emulation library, patch torch.cuda.  Returns the library path."""
    global _STATE
    if _STATE is not None:
        return _STATE["so"]
    import torch
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    import build as hostemu_build
    from qwen3_tts_amd import _lib
    so = hostemu_build.build()
    saved_lib, saved_env = _lib._LIB, os.environ.get("QTTS_LIBRARY")
    os.environ["QTTS_LIBRARY"] = so
    _lib._LIB = None
    patches = [
        mock.patch.object(_lib, "hip_device", lambda device, who: torch.device("cpu")),
        mock.patch.object(torch.cuda, "device", _NullDeviceCtx),
        mock.patch.object(torch.cuda, "current_stream", lambda *a, **k: _FakeStream()),
        mock.patch.object(torch.cuda, "Stream", _FakeStream),
        mock.patch.object(torch.cuda, "stream", lambda s: contextlib.nullcontext()),
        mock.patch.object(torch.cuda, "synchronize", lambda *a, **k: None),
        mock.patch.object(torch.cuda, "set_device", lambda *a, **k: None),
        mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self),
    ]
    for p in patches:
        p.start()
    _STATE = {"so": so, "patches": patches, "saved_lib": saved_lib, "saved_env": saved_env}
    assert os.path.basename(_lib.library_path()).startswith("libqtts_hostemu")
    _lib.load_library()
    return so


def uninstall():
    global _STATE
    if _STATE is None:
        return
    from qwen3_tts_amd import _lib
    for p in reversed(_STATE["patches"]):
        p.stop()
    _lib._LIB = _STATE["saved_lib"]
    if _STATE["saved_env"] is None:
        os.environ.pop("QTTS_LIBRARY", None)
    else:
        os.environ["QTTS_LIBRARY"] = _STATE["saved_env"]
    _STATE = None
