#!/usr/bin/env python3
"""In-kernel phase timestamps (tstamp build variant) of the decode GEMM in hipGraph CHAINS of one launch repeated
(qtts_debug_skinny_chain): operator L2-resident vs streamed (rotating copies).  Next to tools/ts_frame.py's table of the real
frame step this says which phase of a launch is longer there: the wait for the first bytes, or the boundary.
Run with QTTS_LIBRARY=qwen3-tts_amd/libqtts_tstamp.so."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from qwen3_tts_amd import _lib
lib = _lib.load_library()
assert hasattr(lib, "qtts_debug_tslog_skinny"), "needs the tstamp build variant (QTTS_LIBRARY=.../libqtts_tstamp.so)"
torch.zeros(1).cuda()
REC = np.dtype([("t", "<u8", 6), ("kind", "<i4"), ("a", "<i4"), ("b", "<i4"), ("blk", "<i4")])
CAP = 1 << 15
dr = lib.qtts_debug_tslog_skinny; dr.argtypes = [C.c_void_p, C.c_int]; dr.restype = C.c_int
def drain():
    buf = np.zeros(CAP, dtype=REC); n = dr(buf.ctypes.data, CAP); assert n >= 0
    return buf[:n].copy()
f = lib.qtts_debug_skinny_chain
f.argtypes = [C.c_int32] * 9 + [C.POINTER(C.c_double)]; f.restype = C.c_int
def engine_fs(N, K):
    floor = 192 if N * K * 2 >= (16 << 20) else 96
    fs = 16
    while fs > 4 and N // fs < floor: fs //= 2
    return fs
def run(N, K, act, norm, res, wbufs, iters=120):
    _lib.set_option("QTTS_DEBUG_WBUFS", str(wbufs)); _lib.set_option("QTTS_DEBUG_FS", str(engine_fs(N, K)))
    us = C.c_double()
    drain()
    assert f(N, K, 8, act, norm, res, 0, iters, 1, C.byref(us)) == 0, lib.qtts_last_error()
    r = drain()
    r = r[np.argsort(r["t"][:, 0], kind="stable")][-2 * iters:]          # the last (timed) replay: two records (first / last workgroup) per launch
    ph = (r["t"][:, 1:].astype(np.float64) - r["t"][:, :1].astype(np.float64)) * 0.01
    ph[r["t"][:, 1:] == 0] = np.nan
    ent = np.sort(r["t"][:, 0].astype(np.float64))[::2] * 0.01
    end = r["t"].max(axis=1).astype(np.float64) * 0.01
    order = np.argsort(r["t"][:, 0], kind="stable")
    end_l = np.maximum(end[order][0::2], end[order][1::2])
    gaps = ent[1:] - end_l[:-1]
    return us.value, np.nanmean(ph, axis=0), float(np.median(gaps))
print("shape                          buffers  us/launch | issued arrived mfma+lds barrier stored | boundary")
for name, N, K, act, norm, res in (("cp q|k|v 4096x1024 norm", 4096, 1024, 0, 1, 0), ("cp o 1024x2048 res", 1024, 2048, 0, 0, 1),
                                   ("cp gate|up 6144x1024 swiglu", 6144, 1024, 2, 1, 0), ("cp down 1024x3072 res", 1024, 3072, 0, 0, 1),
                                   ("tk o 2048x2048 res", 2048, 2048, 0, 0, 1), ("tk down 2048x6144 res", 2048, 6144, 0, 0, 1)):
    mb = N * K * 2 / 1e6
    for n in (1, max(2, int(400 / mb) + 1)):
        us, ph, gap = run(N, K, act, norm, res, n)
        print(f"{name:30s} {n:7d} {us:10.2f} | " + " ".join(f"{v:6.2f}" for v in ph) + f" | {gap:6.2f}", flush=True)
