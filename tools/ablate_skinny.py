"""Ablation of the decode GEMM (perf tooling): hipGraph chains of identical launches, parts of the kernel disabled."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from qwen3_tts_amd import _lib
# the ablation branches exist only in the `ablate` build variant (python qwen3-tts_amd/build.py --variant ablate)
import os as _os
_os.environ.setdefault("QTTS_LIBRARY", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "qwen3-tts_amd", "libqtts_ablate.so"))
lib = _lib.load_library()
torch.zeros(1).cuda()
f = lib.qtts_debug_skinny_chain
f.argtypes = [C.c_int32] * 9 + [C.POINTER(C.c_double)]; f.restype = C.c_int
def run(N, K, M, act, norm, res, abl, iters=200, reps=5):
    us = C.c_double()
    rc = f(N, K, M, act, norm, res, abl, iters, reps, C.byref(us))
    assert rc == 0, lib.qtts_last_error()
    return us.value
shapes = [("cp_qkv 4096x1024 norm", 4096, 1024, 0, 1, 0), ("cp_o 1024x2048 res", 1024, 2048, 0, 0, 1), ("cp_gu 6144x1024 swiglu", 6144, 1024, 2, 1, 0),
          ("cp_down 1024x3072 res", 1024, 3072, 0, 0, 1), ("tk_qkv 4096x2048 norm", 4096, 2048, 0, 1, 0), ("tk_gu 12288x2048 swiglu", 12288, 2048, 2, 1, 0),
          ("tk_down 2048x6144 res", 2048, 6144, 0, 0, 1)]
abls = [(0, "full"), (1, "-done"), (3, "-done-x"), (7, "-done-x-epi"), (15, "-done-x-epi-w"), (8, "-w only"), (2, "-x only"), (4, "-epi only")]
print("shape".ljust(28) + "".join(n.rjust(15) for _, n in abls) + "   MB   GB/s(full)")
for name, N, K, act, norm, res in shapes:
    r = [run(N, K, 8, act, norm, res, a) for a, _ in abls]
    mb = N * K * 2 / 1e6
    print(name.ljust(28) + "".join(f"{v:15.2f}" for v in r) + f"  {mb:5.1f}  {mb / r[0] * 1e3:7.0f}")
g = lib.qtts_debug_null_chain
g.argtypes = [C.c_int32] * 5 + [C.POINTER(C.c_double)]; g.restype = C.c_int
print("null-kernel chains (us/launch):")
for grid, block in ((1, 64), (256, 64), (256, 256), (256, 512), (512, 512), (1024, 256), (64, 512)):
    r = []
    for mode in (0, 1, 2):
        us = C.c_double(); assert g(grid, block, mode, 400, 5, C.byref(us)) == 0; r.append(us.value)
    print(f"  grid {grid:5d} x block {block:4d}: empty {r[0]:.2f}  +flag-load {r[1]:.2f}  +lds/barrier {r[2]:.2f}")
