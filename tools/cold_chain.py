#!/usr/bin/env python3
"""What does a decode-GEMM launch cost when only ONE thing is cold?  hipGraph chains of the same launch (qtts_debug_skinny_chain):
the operator L2-resident (1 buffer), rotating through 120 MB of copies (inside the 256 MB Infinity Cache) and through 400 MB (from
HBM), same kernel every time.  The
frame step differs from the streamed chain only in that consecutive launches are DIFFERENT kernels (instruction cache, kernel
descriptors) with attention / sampler launches between them."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from qwen3_tts_amd import _lib
lib = _lib.load_library()
torch.zeros(1).cuda()
f = lib.qtts_debug_skinny_chain
f.argtypes = [C.c_int32] * 9 + [C.POINTER(C.c_double)]; f.restype = C.c_int
def engine_fs(N, K):                                     # talker_engine.hip choose_fs
    floor = max(192, 96) if N * K * 2 >= (16 << 20) else 96
    fs = 16
    while fs > 4 and N // fs < floor: fs //= 2
    return fs
def run(N, K, M, act, norm, res, wbufs, iters=240, reps=5):
    _lib.set_option("QTTS_DEBUG_WBUFS", str(wbufs)); _lib.set_option("QTTS_DEBUG_FS", str(engine_fs(N, K)))
    us = C.c_double()
    rc = f(N, K, M, act, norm, res, 0, iters, reps, C.byref(us))
    assert rc == 0, lib.qtts_last_error()
    return us.value
print("shape (M = 8)                 MB   resident us   120 MB rotation us   400 MB rotation us   GB/s (400 MB)")
for name, N, K, act, norm, res in (("cp q|k|v 4096x1024 norm", 4096, 1024, 0, 1, 0), ("cp o 1024x2048 res", 1024, 2048, 0, 0, 1),
                                   ("cp gate|up 6144x1024 swiglu", 6144, 1024, 2, 1, 0), ("cp down 1024x3072 res", 1024, 3072, 0, 0, 1),
                                   ("tk q|k|v 4096x2048 norm", 4096, 2048, 0, 1, 0), ("tk o 2048x2048 res", 2048, 2048, 0, 0, 1),
                                   ("tk gate|up 12288x2048 swiglu", 12288, 2048, 2, 1, 0), ("tk down 2048x6144 res", 2048, 6144, 0, 0, 1)):
    mb = N * K * 2 / 1e6
    n = max(2, int(400 / mb) + 1)
    m = max(2, int(120 / mb))
    a, c, b = run(N, K, 8, act, norm, res, 1), run(N, K, 8, act, norm, res, m), run(N, K, 8, act, norm, res, n)
    print(f"{name:28s} {mb:5.1f} {a:12.2f} {c:20.2f} {b:20.2f} {mb / b * 1e3:16.0f}", flush=True)
