"""TEST INFRASTRUCTURE: build tests/hostemu/libqtts_hostemu.so -- the product's C++ / HIP sources compiled for the HOST.

  * engines (csrc/*_engine.hip): plain host C++ against hip/hip_runtime.h (device memory = host memory);
  * kernels (SIMT_KERNELS = every kernel source of csrc/): their REAL sources, executed by the SIMT emulator of simt.h
    (one fiber per thread, wave collectives, workgroup barriers, MFMA / DPP / LDS-DMA semantics) -- the only edit is
    mechanical: `extern __shared__` -> `extern` (dynamic LDS is a host array, lds_arrays.cpp), `__shared__` -> `static`;
  * one stand-in (cpu_gemm_tap.cpp): the tap GEMM in plain loops, used by the large engine tests because the MFMA
    emulation is ~15x slower; the real gemm_tap.hip is in the library too (launch_gemm_tap_real) and is selected by
    hostemu_set_real_gemm(1) / QTTS_HOSTEMU_FULL=1.
The library exports the whole C ABI on host pointers; tests/test_hostemu.py drives it with numpy arrays."""
import hashlib
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "qwen3-tts_amd", "csrc")
# QTTS_HOSTEMU_ASAN=1: an AddressSanitizer build (libqtts_hostemu_asan.so): heap redzones around every device buffer,
# global redzones around every static LDS array.  Run it as
#   LD_PRELOAD=$(python tests/hostemu/build.py --asan-runtime) ASAN_OPTIONS=detect_leaks=0 QTTS_HOSTEMU_ASAN=1 \
#       python -m pytest tests/test_hostemu.py
ASAN = os.environ.get("QTTS_HOSTEMU_ASAN") == "1"
# QTTS_HOSTEMU_UBSAN=1: an UndefinedBehaviorSanitizer build (libqtts_hostemu_ubsan.so), trapping on the first finding:
# misaligned vector accesses (float4 / uint4 / uint2 are alignas(16 / 8) in the stand-in hip_runtime.h, as the hardware's
# ds_read_b128 / dwordx4 paths want them), signed overflow in index arithmetic, out-of-range shifts, division by zero,
# out-of-bounds indexing of fixed-size arrays.  No runtime library needed (-fsanitize-trap): a finding is a SIGILL.
UBSAN = os.environ.get("QTTS_HOSTEMU_UBSAN") == "1" and not ASAN
# QTTS_HOSTEMU_DEFS="-DQTTS_ATTN_TAIL_BATCH=1 ...": the same emulated library with a build variant's flags
# (qwen3-tts_amd/build.py VARIANTS), so that a variant is checked against the oracle before it ever meets hardware.
DEFS = os.environ.get("QTTS_HOSTEMU_DEFS", "").split()
_TAG = ("_asan" if ASAN else "_ubsan" if UBSAN else "") + ("_v" + hashlib.sha256(" ".join(DEFS).encode()).hexdigest()[:8] if DEFS else "")
OUT = os.path.join(HERE, f"libqtts_hostemu{_TAG}.so")
GEN = os.path.join(HERE, "gen" + _TAG)
ENGINES = ["codec_engine.hip", "encoder_engine.hip", "speaker_engine.hip", "talker_engine.hip"]
SIMT_KERNELS = ["stream_kernels.hip", "encoder_kernels.hip", "speaker_kernels.hip", "attention.hip", "cp_mlp.hip", "cp_mlp32.hip", "cp_layer.hip", "sampling.hip",
                "elementwise.hip", "skinny.hip", "gemm_tap.hip", "resunit.hip"]
# kernels without barriers / cross-lane ops run as plain per-thread calls (no fibers): much faster for large grids
SEQUENTIAL = {"stream_kernels.hip", "speaker_kernels.hip"}
# gemm_tap.hip is built as launch_gemm_tap_real; cpu_gemm_tap.cpp owns launch_gemm_tap and forwards to it on request
EXTRA_DEFS = {"gemm_tap.hip": ["-Dlaunch_gemm_tap=launch_gemm_tap_real"]}
STANDIN = ["cpu_gemm_tap.cpp", "test_entries.cpp", "lds_arrays.cpp"]
HEADERS = [os.path.join(CSRC, h) for h in ("common.h", "kernels.h", "glue.h", "granule.h", "attn_helpers.h", "tstamp.h")] + [
    os.path.join(ROOT, "include", "qtts.h"), os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "hip", "hip_ext.h"), os.path.join(HERE, "simt.h")]


def _compiler():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/bin/amdclang++", "clang++"):   # ext_vector_type needs clang
        if os.path.exists(c) or c == "clang++":
            return c


def _transform(src_path, dst_path):
    s = open(src_path).read()
    s = s.replace("extern __shared__", "extern")
    s = re.sub(r"\b__shared__\b", "static", s)
    with open(dst_path, "w") as f:
        f.write(f'#line 1 "{src_path}"\n' + s)


def build(verbose=False):
    inputs = [os.path.join(CSRC, f) for f in ENGINES + SIMT_KERNELS] + [os.path.join(HERE, f) for f in STANDIN] + HEADERS + [__file__]
    h = hashlib.sha256()
    for f in inputs:
        with open(f, "rb") as fh:
            h.update(fh.read())
    stamp = OUT + ".sha"
    if os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return OUT
    os.makedirs(GEN, exist_ok=True)
    cc = _compiler()
    base = [cc, "-std=c++17", "-O2", "-fPIC", "-DQTTS_HOST_EMU", "-I", HERE, "-I", CSRC, "-Wno-unused-function", "-Wno-unused-value"]
    san = ["-fsanitize=address", "-shared-libasan", "-fno-omit-frame-pointer", "-g"] if ASAN else []
    if UBSAN:
        checks = "alignment,signed-integer-overflow,shift,integer-divide-by-zero,bounds,null"
        san = [f"-fsanitize={checks}", f"-fsanitize-trap={checks}", "-fno-omit-frame-pointer", "-g"]
    base += san + DEFS
    objs = []
    for f in ENGINES + SIMT_KERNELS:
        dst = os.path.join(GEN, f.replace(".hip", ".cpp"))
        _transform(os.path.join(CSRC, f), dst)
        objs.append((dst, (["-DQTTS_SIMT_SEQUENTIAL"] if f in SEQUENTIAL else []) + EXTRA_DEFS.get(f, [])))
    objs += [(os.path.join(HERE, f), []) for f in STANDIN]
    outs = []
    for src, extra in objs:
        o = os.path.join(GEN, os.path.basename(src) + ".o")
        cmd = base + extra + ["-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        outs.append(o)
    subprocess.run([cc, "-shared", "-o", OUT] + san + outs, check=True)
    with open(stamp, "w") as fh:
        fh.write(h.hexdigest())
    return OUT


def asan_runtime():
    out = subprocess.run([_compiler(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True, check=True)
    return out.stdout.strip()


if __name__ == "__main__":
    import sys
    print(asan_runtime() if "--asan-runtime" in sys.argv else build(verbose=True))
