#!/usr/bin/env python3
"""Microbenchmark of the small-grid GEMM (gemm_wide_kernel via launch_gemm_tap, qtts_debug_gemm_tap): the talker prefill's and the
codec transformer's shapes, per tile (qtts_debug_gemm_wide_tile; a tile the shape or the registers do not admit runs as 128 x 128) and with fp32 / bf16 activations.  us per launch in a hipGraph chain."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
SHAPES = [("prefill q|k|v", 512, 4096, 2048, 0, 0), ("prefill o", 512, 2048, 2048, 0, 1), ("prefill gate|up", 512, 12288, 2048, 2, 0),
          ("prefill down", 512, 2048, 6144, 0, 1),
          ("prefill b32 q|k|v", 2048, 4096, 2048, 0, 0), ("prefill b32 gate|up", 2048, 12288, 2048, 2, 0), ("prefill b32 down", 2048, 2048, 6144, 0, 1),
          ("codec q|k|v", 1000, 3072, 1024, 0, 0), ("codec o", 1000, 1024, 1024, 0, 1), ("codec gate|up", 1000, 6144, 1024, 2, 0),
          ("codec down", 1000, 1024, 3072, 0, 1), ("codec b1 q|k|v", 125, 3072, 1024, 0, 0), ("codec b1 o", 125, 1024, 1024, 0, 1), ("codec b1 gate|up", 125, 6144, 1024, 2, 0), ("codec b1 down", 125, 1024, 3072, 0, 1),
          ("prefill 1x40 q|k|v", 40, 4096, 2048, 0, 0), ("prefill 1x40 o", 40, 2048, 2048, 0, 1), ("prefill 1x40 gate|up", 40, 12288, 2048, 2, 0), ("prefill 1x40 down", 40, 2048, 6144, 0, 1)]
if "--small" in sys.argv: SHAPES = [s for s in SHAPES if s[1] <= 128]
import torch
from qwen3_tts_amd import _lib
lib = _lib.load_library()
torch.zeros(1).cuda()
f = lib.qtts_debug_gemm_tap
f.argtypes = [C.c_int32] * 8 + [C.POINTER(C.c_double)]; f.restype = C.c_int
lib.qtts_debug_gemm_wide_tile.argtypes = [C.c_int32]; lib.qtts_debug_gemm_wide_tile.restype = None
cols = [("chooser", 0), ("128x128", 128128128), ("128x64", 128064128), ("64x128", 64128128), ("64x64", 64064128), ("64x128 k256", 64128256), ("64x64 k256", 64064256), ("32x64 k256", 32064256), ("32x32 k256", 32032256)]
print(f"{'shape (M x N x K)':44s} A    " + "".join(f"{c:>13s}" for c, _ in cols))
for name, M, N, K, act, rs in SHAPES:
    for a16 in (0, 1):
        vals = []
        for cname, code in cols:
            lib.qtts_debug_gemm_wide_tile(code)
            us = C.c_double()
            rc = f(M, N, K, act, rs, a16, 40, 4, C.byref(us))
            vals.append(us.value if rc == 0 else None)
        print(f"{name:20s} {M:5d} x {N:5d} x {K:5d}  {'bf16' if a16 else 'fp32'} " + "".join(f"{v:13.2f}" if v else f"{'-':>13s}" for v in vals), flush=True)
