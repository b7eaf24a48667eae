"""Build libqtts.so (the C-ABI HIP library) in-tree for gfx950.

    python qwen3-tts_amd/build.py [--force]

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the tree.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libqtts.so")
SOURCES = ["gemm_tap.hip", "skinny.hip", "elementwise.hip", "attention.hip", "sampling.hip",
           "codec_engine.hip", "talker_engine.hip", "encoder_kernels.hip", "encoder_engine.hip",
           "speaker_kernels.hip", "speaker_engine.hip", "stream_kernels.hip"]
HEADERS = ["common.h", "kernels.h", "glue.h", os.path.join("..", "..", "include", "qtts.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        stamp = obj + ".sha"
        dig = _digest([src] + hdrs)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), stamp, dig, s))
    failed = []
    for p, stamp, dig, s in procs:
        if p.wait() != 0:
            failed.append(s)
        else:
            with open(stamp, "w") as f:
                f.write(dig)
    if failed:
        raise RuntimeError(f"hipcc failed for: {failed}")
    if procs or not os.path.exists(OUT) or force:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
