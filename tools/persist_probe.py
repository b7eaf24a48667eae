#!/usr/bin/env python3
"""Persistent launch with grid barriers vs a hipGraph chain of launches, on the GEMM chain of one code-predictor layer
(csrc/persist_probe.hip; VERDICT r2 item 3d: decide by measurement, not by the price list).  Prints microseconds per GEMM stage
for both sides, for 1, 2 and 5 layers (4, 8, 20 stages), and how far the two sides' outputs are apart."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
# the probe lives in its own build variant (python qwen3-tts_amd/build.py --variant probe), not in the product library
os.environ.setdefault("QTTS_LIBRARY", os.path.join(ROOT, "qwen3-tts_amd", "libqtts_probe.so"))
os.environ.setdefault("QTTS_LIBRARY_OK", "1")
from qwen3_tts_amd import _lib
lib = _lib.load_library()
torch.zeros(1).cuda()
f = lib.qtts_debug_persist_layer
f.argtypes = [C.c_int32, C.c_int32] + [C.POINTER(C.c_double)] * 3 + [C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]; f.restype = C.c_int
print("layers stages | persistent us/stage | launches us/stage | ratio | launches + prefetch branch | max rel diff | barrier gave up | workgroup 0, us per stage: body  arrive  request  wait")
for L in (1, 2, 5):
    for rep in range(2):
        a, b, d, ab, ph, pf = C.c_double(), C.c_double(), C.c_double(), C.c_int32(), (C.c_double * 4)(), C.c_double()
        rc = f(L, 20, C.byref(a), C.byref(b), C.byref(d), C.byref(ab), ph, C.byref(pf))
        assert rc == 0, lib.qtts_last_error()
        print(f"{L:6d} {4 * L:6d} | {a.value:19.2f} | {b.value:17.2f} | {a.value / b.value:5.2f} | {pf.value:26.2f} | {d.value:12.4f} | {ab.value} | " + "  ".join(f"{v:.2f}" for v in ph), flush=True)
