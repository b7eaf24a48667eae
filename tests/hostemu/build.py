"""TEST INFRASTRUCTURE: build tests/hostemu/libqtts_hostemu.so = the two engine orchestration files of the product
(csrc/codec_engine.hip, csrc/encoder_engine.hip, csrc/speaker_engine.hip) compiled as HOST C++ against tests/hostemu/hip/hip_runtime.h, linked with
CPU versions of the kernel launch interfaces (cpu_kernels.cpp).  The library exports the codec + encoder part of the C ABI
on host pointers; tests/test_hostemu.py drives it with numpy arrays and checks it against the oracle."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "qwen3-tts_amd", "csrc")
OUT = os.path.join(HERE, "libqtts_hostemu.so")
SRCS = [os.path.join(CSRC, "codec_engine.hip"), os.path.join(CSRC, "encoder_engine.hip"),
        os.path.join(CSRC, "speaker_engine.hip"),
        # thread-independent kernels: the REAL sources, run by the sequential interpreter of hip/hip_runtime.h
        os.path.join(CSRC, "stream_kernels.hip"), os.path.join(CSRC, "encoder_kernels.hip"), os.path.join(CSRC, "speaker_kernels.hip"),
        os.path.join(CSRC, "talker_engine.hip"), os.path.join(HERE, "cpu_talker_kernels.cpp"),
        os.path.join(HERE, "cpu_kernels.cpp")]
DEPS = SRCS + [os.path.join(CSRC, h) for h in ("common.h", "kernels.h", "glue.h")] + [os.path.join(ROOT, "include", "qtts.h"),
                                                                            os.path.join(HERE, "hip", "hip_runtime.h")]


def _compiler():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/bin/amdclang++", "clang++"):   # ext_vector_type needs clang
        if os.path.exists(c) or c == "clang++":
            return c


def build(verbose=False):
    h = hashlib.sha256()
    for f in DEPS:
        with open(f, "rb") as fh:
            h.update(fh.read())
    stamp = OUT + ".sha"
    if os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return OUT
    cmd = [_compiler(), "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-DQTTS_HOST_EMU", "-I", HERE, "-I", CSRC,
           "-Wno-unused-function", "-o", OUT] + SRCS
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as fh:
        fh.write(h.hexdigest())
    return OUT


if __name__ == "__main__":
    print(build(verbose=True))
