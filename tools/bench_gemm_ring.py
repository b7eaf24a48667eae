#!/usr/bin/env python3
"""A/B of the ring tap GEMM (gemm_ring_kernel, round 6) against gemm_dma_kernel on the codec decoder's convolution shapes (C = 768 / 384 units at
1 x 10 s and 8 x 10 s; the two-tap transposed form) and on the prefill's plain Linears against gemm_wide, interleaved in one process
(qtts_debug_gemm_tap16 / qtts_debug_gemm_tap: a hipGraph chain of launches, best of 4).  Every ring output is compared BITWISE with gemm_dma's
(same MFMA sequence per accumulator), `--screen N` repeats that comparison N times per shape (race screen).  TFLOP/s = 2 M N K taps / time."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
ap = argparse.ArgumentParser(); ap.add_argument("--screen", type=int, default=3); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--skip-linear", action="store_true")
ap.add_argument("--ablate", action="store_true", help="time the measuring variants of the 4-deep 7-tap kernel (QTTS_GEMM_RING_ABLATE; they exist only in the `ablate` build "
                "variant: python qwen3-tts_amd/build.py --variant ablate) on three shapes and exit")
a = ap.parse_args()
if a.ablate: os.environ.setdefault("QTTS_LIBRARY", os.path.join(ROOT, "qwen3-tts_amd", "libqtts_ablate.so"))
from qwen3_tts_amd import _lib
lib = _lib.load_library(); torch.zeros(1).cuda()
f16 = lib.qtts_debug_gemm_tap16
f16.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
f16.restype = C.c_int
g = np.random.default_rng(6)
def bf16bits(x): return ((x.view(np.uint32) + 0x7fff + ((x.view(np.uint32) >> 16) & 1)) >> 16).astype(np.uint16)
def run(A, W, T, shifts, iters):
    M, K = A.shape; taps, N, _ = W.shape
    out = np.empty((M, N), np.float32); us = C.c_double(0)
    sh = (C.c_int32 * taps)(*shifts)
    rc = f16(A.ctypes.data, K, M, T, W.ctypes.data, N, K, taps, sh, out.ctypes.data, iters, 4 if iters else 0, C.byref(us))
    assert rc == 0, lib.qtts_last_error()
    return out, us.value
VARIANTS = [("dma", {"QTTS_GEMM_RING": "0"}), ("ring4", {"QTTS_GEMM_RING": "2", "QTTS_GEMM_RING_NST": "4", "QTTS_GEMM_RING_KS": "1"}),
            ("ring6", {"QTTS_GEMM_RING": "2", "QTTS_GEMM_RING_NST": "6", "QTTS_GEMM_RING_KS": "1"}), ("ring8", {"QTTS_GEMM_RING": "2", "QTTS_GEMM_RING_NST": "8", "QTTS_GEMM_RING_KS": "1"})]
conv7 = lambda d: [-(6 - j) * d for j in range(7)]
SHAPES = []
for B in (1, 8):
    SHAPES += [(f"C768 conv7 d1 B{B}", 4000 * B, 4000, 768, 768, conv7(1)), (f"C768 conv7 d9 B{B}", 4000 * B, 4000, 768, 768, conv7(9)),
               (f"C384 conv7 d3 B{B}", 20000 * B, 20000, 384, 384, conv7(3)), (f"C384 conv7 d9 B{B}", 20000 * B, 20000, 384, 384, conv7(9)),
               (f"C384 conv1 B{B}", 20000 * B, 20000, 384, 384, [0]), (f"tconv 768->5x384 B{B}", 4000 * B, 4000, 1920, 768, [0, -1]),
               (f"tconv 1536->8x768 B{B}", 500 * B, 500, 6144, 1536, [0, -1])]
SHAPES.append(("32 x 4-frame C768 d9", 32 * 128, 128, 768, 768, conv7(9)))
if a.ablate:
    AB = [(0, "all"), (1, "no requests"), (2, "no fragment reads"), (16, "no A address arithmetic"), (4, "no MFMAs"), (8, "no barrier"), (3, "no requests, no reads"), (7, "no requests / reads / MFMAs"), (15, "nothing but the cursors and waits")]
    for name, M, T, N, K, shifts in [SHAPES[0], SHAPES[7], SHAPES[8]]:
        A = bf16bits((g.standard_normal((M, K), dtype=np.float32) * 0.5)); W = bf16bits((g.standard_normal((len(shifts), N, K), dtype=np.float32) / np.sqrt(K * len(shifts))).astype(np.float32))
        steps = K // 32 * len(shifts)
        print(f"{name}: {M} x {N} x {K} x {len(shifts)} taps, {steps} steps per workgroup, {((M + 127) // 128) * (N // 128)} workgroups")
        for code, what in AB:
            best = 1e30
            for rep in range(2):
                with _lib.options(QTTS_GEMM_RING="1", QTTS_GEMM_RING_NST="4", QTTS_GEMM_RING_ABLATE=str(code)):
                    best = min(best, run(A, W, T, shifts, a.iters)[1])
            print(f"  ablate {code:2d} ({what:36s}) {best:8.2f} us", flush=True)
    sys.exit(0)
print(f"{'shape':28s} {'M':>7s} {'N':>5s} {'K':>5s} taps " + " ".join(f"{n + ' us':>10s} {'TF/s':>6s}" for n, _ in VARIANTS) + "  bitwise")
bad = 0
for name, M, T, N, K, shifts in SHAPES:
    A = bf16bits((g.standard_normal((M, K), dtype=np.float32) * 0.5)); W = bf16bits((g.standard_normal((len(shifts), N, K), dtype=np.float32) / np.sqrt(K * len(shifts))).astype(np.float32))
    res = {}; ref = None; same = True
    for rep in range(2):
        for vn, opts in VARIANTS:
            with _lib.options(**opts):
                out, us = run(A, W, T, shifts, a.iters)
            res[vn] = min(res.get(vn, 1e30), us)
            if ref is None: ref = out
            elif not np.array_equal(out, ref): same = False
    for _ in range(a.screen):
        for vn, opts in VARIANTS[1:]:
            with _lib.options(**opts):
                out, _u = run(A, W, T, shifts, 0)
            if not np.array_equal(out, ref): same = False
    bad += 0 if same else 1
    fl = 2.0 * M * N * K * len(shifts)
    print(f"{name:28s} {M:7d} {N:5d} {K:5d} {len(shifts):4d} " + " ".join(f"{res[n]:10.2f} {fl / res[n] / 1e6:6.0f}" for n, _ in VARIANTS) + ("  identical" if same else "  DIFFERENT"), flush=True)
print("bitwise mismatches:", bad)
# ---- split-K (ordered combine behind a ticket): time per launch at 1 .. 4 splits, and the default rule
KSV = [("ks1", "1"), ("ks2", "2"), ("ks3", "3"), ("ks4", "4"), ("default", "0")]
print(f"\n{'split-K, convolution form':28s} {'M':>7s} {'N':>5s} {'K':>5s} taps " + " ".join(f"{n + ' us':>10s}" for n, _ in KSV))
for name, M, T, N, K, shifts in [SHAPES[0], SHAPES[1], SHAPES[2], SHAPES[5], SHAPES[6], SHAPES[14]]:
    A = bf16bits((g.standard_normal((M, K), dtype=np.float32) * 0.5)); W = bf16bits((g.standard_normal((len(shifts), N, K), dtype=np.float32) / np.sqrt(K * len(shifts))).astype(np.float32))
    r = {}
    for rep in range(2):
        for vn, ks in KSV:
            with _lib.options(QTTS_GEMM_RING="1", QTTS_GEMM_RING_KS=ks):
                r[vn] = min(r.get(vn, 1e30), run(A, W, T, shifts, a.iters)[1])
    print(f"{name:28s} {M:7d} {N:5d} {K:5d} {len(shifts):4d} " + " ".join(f"{r[n]:10.2f}" for n, _ in KSV), flush=True)
if not a.skip_linear:
    fl2 = lib.qtts_debug_gemm_tap
    fl2.argtypes = [C.c_int32] * 8 + [C.POINTER(C.c_double)]; fl2.restype = C.c_int
    LIN2 = [("b32 o (1536 rows)", 1536, 2048, 2048, 0, 1), ("b32 down (1536 rows)", 1536, 2048, 6144, 0, 1), ("b32 q|k|v (1536 rows)", 1536, 4096, 2048, 0, 0),
            ("b8 o (448 rows)", 448, 2048, 2048, 0, 1), ("b8 down (448 rows)", 448, 2048, 6144, 0, 1), ("b8 q|k|v (448 rows)", 448, 4096, 2048, 0, 0),
            ("b8 o (512 rows)", 512, 2048, 2048, 0, 1), ("b8 down (512 rows)", 512, 2048, 6144, 0, 1)]
    print(f"\n{'split-K, plain Linear (M x N x K)':46s} {'wide us':>10s} " + " ".join(f"{n + ' us':>10s}" for n, _ in KSV))
    for name, M, N, K, act, rs in LIN2:
        r = {}
        for rep in range(2):
            with _lib.options(QTTS_GEMM_RING="0", QTTS_GEMM_DMA="0"):
                us = C.c_double(); assert fl2(M, N, K, act, rs, 1, 40, 4, C.byref(us)) == 0; r["wide"] = min(r.get("wide", 1e9), us.value)
            for vn, ks in KSV:
                with _lib.options(QTTS_GEMM_RING=("1" if vn == "default" else "2"), QTTS_GEMM_RING_KS=ks):
                    us = C.c_double(); assert fl2(M, N, K, act, rs, 1, 40, 4, C.byref(us)) == 0; r[vn] = min(r.get(vn, 1e9), us.value)
        print(f"{name:24s} {M:5d} x {N:5d} x {K:5d}   {r['wide']:10.2f} " + " ".join(f"{r[n]:10.2f}" for n, _ in KSV), flush=True)
if not a.skip_linear:
    fl_ = lib.qtts_debug_gemm_tap
    fl_.argtypes = [C.c_int32] * 8 + [C.POINTER(C.c_double)]; fl_.restype = C.c_int
    LIN = [("prefill b32 q|k|v", 2048, 4096, 2048, 0, 0), ("prefill b32 o", 2048, 2048, 2048, 0, 1), ("prefill b32 gate|up", 2048, 12288, 2048, 2, 0),
           ("prefill b32 down", 2048, 2048, 6144, 0, 1), ("prefill b8 q|k|v", 512, 4096, 2048, 0, 0), ("prefill b8 gate|up", 512, 12288, 2048, 2, 0),
           ("prefill b8 down", 512, 2048, 6144, 0, 1), ("square 4096", 4096, 4096, 4096, 0, 0)]
    LV = [("wide", {"QTTS_GEMM_RING": "0", "QTTS_GEMM_DMA": "0"}), ("dma", {"QTTS_GEMM_RING": "0", "QTTS_GEMM_DMA": "2"}),
          ("ring4", {"QTTS_GEMM_RING": "2", "QTTS_GEMM_RING_NST": "4"}), ("default", {})]
    print(f"\n{'plain Linear (M x N x K)':46s} " + " ".join(f"{n + ' us':>10s} {'TF/s':>6s}" for n, _ in LV))
    for name, M, N, K, act, rs in LIN:
        r = {}
        for rep in range(2):
            for vn, opts in LV:
                with _lib.options(**opts):
                    us = C.c_double(); rc = fl_(M, N, K, act, rs, 1, 40, 4, C.byref(us)); assert rc == 0, lib.qtts_last_error()
                r[vn] = min(r.get(vn, 1e9), us.value)
        fl = 2.0 * M * N * K
        print(f"{name:20s} {M:5d} x {N:5d} x {K:5d}   " + " ".join(f"{r[n]:10.2f} {fl / r[n] / 1e6:6.0f}" for n, _ in LV), flush=True)
sys.exit(1 if bad else 0)
