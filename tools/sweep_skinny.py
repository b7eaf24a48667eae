#!/usr/bin/env python3
"""Microbenchmark sweep of the bf16 decode GEMM (hipGraph chains of identical launches, qtts_debug_skinny_chain): strip width
(= number of workgroups), batch rows, weight-load flavour.  What scales the fixed cost of a launch?"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from qwen3_tts_amd import _lib
lib = _lib.load_library()
torch.zeros(1).cuda()
f = lib.qtts_debug_skinny_chain
f.argtypes = [C.c_int32] * 9 + [C.POINTER(C.c_double)]; f.restype = C.c_int
def run(N, K, M, act=0, norm=0, res=1, abl=0, fs=0, temporal=0, iters=200, reps=5):
    _lib.set_option("QTTS_DEBUG_FS", str(fs)); _lib.set_option("QTTS_CP_TEMPORAL", str(temporal))
    us = C.c_double()
    rc = f(N, K, M, act, norm, res, abl, iters, reps, C.byref(us))
    assert rc == 0, lib.qtts_last_error()
    return us.value
print("N x K            M   " + "".join(f"fs={fs:<2d}({'T' if t else 'nt'}) ".rjust(12) for fs in (4, 8, 16) for t in (0, 1)))
for (N, K) in ((1024, 1024), (1024, 2048), (1024, 3072), (4096, 1024), (2048, 2048), (2048, 6144)):
    for M in (1, 8, 16):
        r = [run(N, K, M, fs=fs, temporal=t) for fs in (4, 8, 16) for t in (0, 1)]
        print(f"{N:5d} x {K:5d}  {M:3d}   " + "".join(f"{v:12.2f}" for v in r), flush=True)
print("gate/up (strip pairs), norm:")
for (N, K) in ((6144, 1024), (12288, 2048)):
    for M in (1, 8, 16):
        r = [run(N, K, M, act=2, norm=1, res=0, temporal=t) for t in (0, 1)]
        print(f"{N:5d} x {K:5d}  {M:3d}   nt {r[0]:.2f}  temporal {r[1]:.2f}", flush=True)
