#!/usr/bin/env python3
"""Race screen of gemm_ring_kernel's counted waits under memory contention: while a second stream keeps the HBM busy with large device-to-device copies
(the latency of the LDS-DMA requests moves), every shape is run `--runs` times through the ring kernel and compared BITWISE with gemm_dma_kernel's
output of the same operands (same MFMA sequence per accumulator).  A tile read before its request landed, or a buffer re-requested under a late reader,
shows as a differing output.  Exit code 1 on any mismatch."""
import argparse, ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from qwen3_tts_amd import _lib
ap = argparse.ArgumentParser(); ap.add_argument("--runs", type=int, default=200); ap.add_argument("--no-contention", action="store_true")
a = ap.parse_args()
lib = _lib.load_library(); torch.zeros(1).cuda()
f16 = lib.qtts_debug_gemm_tap16
f16.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
f16.restype = C.c_int
g = np.random.default_rng(7)
bits = lambda x: ((x.view(np.uint32) + 0x7fff + ((x.view(np.uint32) >> 16) & 1)) >> 16).astype(np.uint16)
def run(A, W, T, shifts):
    M, K = A.shape; taps, N, _ = W.shape
    out = np.empty((M, N), np.float32)
    rc = f16(A.ctypes.data, K, M, T, W.ctypes.data, N, K, taps, (C.c_int32 * taps)(*shifts), out.ctypes.data, 0, 0, None)
    assert rc == 0, lib.qtts_last_error()
    return out
stop = False
def hammer():
    st = torch.cuda.Stream()
    x = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"); y = torch.empty_like(x)       # 256 MiB each: past the Infinity Cache
    with torch.cuda.stream(st):
        while not stop:
            for _ in range(8): y.copy_(x, non_blocking=True)
            st.synchronize()
th = None
if not a.no_contention:
    th = threading.Thread(target=hammer, daemon=True); th.start(); time.sleep(0.5)
conv7 = lambda d: [-(6 - j) * d for j in range(7)]
shapes = [("C768 conv7 d1, 4 x 1000", 4000, 1000, 768, 768, conv7(1)), ("C768 conv7 d9, 8 x 500", 4000, 500, 768, 768, conv7(9)), ("C384 conv7 d3, 16 x 700", 11200, 700, 384, 384, conv7(3)),
          ("tconv 768 -> 5 x 384", 3000, 1500, 1920, 768, [0, -1]), ("Linear 1536 x 2048 x 2048", 1536, 1536, 2048, 2048, [0]), ("Linear 512 x 4096 x 2048", 512, 512, 4096, 2048, [0]),
          ("tiny conv7 d1 (1 slab)", 300, 100, 128, 64, conv7(1)), ("tiny conv7 d3 (3 slabs)", 260, 130, 128, 192, conv7(3))]
bad = 0
for name, M, T, N, K, shifts in shapes:
    A = bits(g.standard_normal((M, K), dtype=np.float32) * 0.5); W = bits((g.standard_normal((len(shifts), N, K), dtype=np.float32) / np.sqrt(K * len(shifts))).astype(np.float32))
    with _lib.options(QTTS_GEMM_RING="0", QTTS_GEMM_DMA="1"):
        ref = run(A, W, T, shifts)
    t0 = time.time(); mis = 0
    for nst in ("4", "8"):
        with _lib.options(QTTS_GEMM_RING="2", QTTS_GEMM_RING_NST=nst, QTTS_GEMM_RING_KS="1"):
            for r in range(a.runs):
                if not np.array_equal(run(A, W, T, shifts), ref): mis += 1
    bad += mis
    print(f"{name:28s} {M:6d} x {N:5d} x {K:5d} x {len(shifts)} taps: {2 * a.runs} runs, {mis} differing ({time.time() - t0:.1f} s)", flush=True)
stop = True
if th is not None: th.join(timeout=30)
print("contention:", "off" if a.no_contention else "256 MiB device-to-device copies on a second stream throughout", "| mismatching runs:", bad)
sys.exit(1 if bad else 0)
