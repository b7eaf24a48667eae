#!/usr/bin/env python3
"""A/B of the LDS-DMA GEMM (gemm_dma_kernel, round 4) against the register-staged small-grid GEMM (gemm_wide_kernel) on the plain
bf16 Linear shapes of the batch-32 and batch-8 prefill, interleaved in one process (qtts_debug_gemm_tap: a hipGraph chain of 40
launches, best of 4); TFLOP/s = 2 M N K / time."""
import os
import ctypes as C, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
SHAPES = [("prefill b32 q|k|v", 2048, 4096, 2048, 0, 0), ("prefill b32 o", 2048, 2048, 2048, 0, 1), ("prefill b32 gate|up", 2048, 12288, 2048, 2, 0),
          ("prefill b32 down", 2048, 2048, 6144, 0, 1), ("prefill b8 q|k|v", 512, 4096, 2048, 0, 0), ("prefill b8 gate|up", 512, 12288, 2048, 2, 0),
          ("prefill b8 down", 512, 2048, 6144, 0, 1), ("square 4096", 4096, 4096, 4096, 0, 0)]
import torch
from qwen3_tts_amd import _lib
lib = _lib.load_library()
torch.zeros(1).cuda()
f = lib.qtts_debug_gemm_tap
f.argtypes = [C.c_int32] * 8 + [C.POINTER(C.c_double)]; f.restype = C.c_int
print(f"{'shape (M x N x K)':46s} {'wide us':>9s} {'TF/s':>7s} {'dma us':>9s} {'TF/s':>7s} {'dma/wide':>9s}")
for name, M, N, K, act, rs in SHAPES:
    r = {}
    for rep in range(2):
        for mode in ("0", "2"):
            _lib.set_option("QTTS_GEMM_DMA", mode)
            us = C.c_double()
            rc = f(M, N, K, act, rs, 1, 40, 4, C.byref(us))
            assert rc == 0, lib.qtts_last_error()
            r[mode] = min(r.get(mode, 1e9), us.value)
    fl = 2.0 * M * N * K
    print(f"{name:20s} {M:5d} x {N:5d} x {K:5d}   {r['0']:9.2f} {fl / r['0'] / 1e6:7.0f} {r['2']:9.2f} {fl / r['2'] / 1e6:7.0f} {r['2'] / r['0']:9.3f}", flush=True)
