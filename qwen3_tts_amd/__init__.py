"""Import alias: the source tree lives in `qwen3-tts_amd/` (hyphen, per the repo layout contract),
which is not an importable name; this package re-exports it as `qwen3_tts_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "qwen3-tts_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
