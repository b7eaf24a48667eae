#!/usr/bin/env bash
# Round 3, GPU call 24: would the talker's gate|up (12288 x 2048, 384 workgroups of a gate / up strip pair) stream faster as 768
# workgroups of one 16-feature strip (three per CU instead of 1.5)?  Same operator bytes, plain epilogue vs SwiGLU epilogue; chains of
# one launch (tools/cold_chain.py machinery), operator resident and streamed.  Also the code predictor's gate|up (6144 x 1024: 192 vs 384).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3w
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
timeout 200 python - > "$OUT/gu_wgs.log" 2>&1 <<'EOF'
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from qwen3_tts_amd import _lib
lib = _lib.load_library()
torch.zeros(1).cuda()
f = lib.qtts_debug_skinny_chain
f.argtypes = [C.c_int32] * 9 + [C.POINTER(C.c_double)]; f.restype = C.c_int
def run(N, K, act, norm, wbufs, fs):
    os.environ["QTTS_DEBUG_WBUFS"] = str(wbufs); os.environ["QTTS_DEBUG_FS"] = str(fs)
    us = C.c_double()
    assert f(N, K, 8, act, norm, 0, 0, 240, 5, C.byref(us)) == 0, lib.qtts_last_error()
    return us.value
print("operator            form                                   resident us   streamed us")
for N, K in ((12288, 2048), (6144, 1024)):
    n = max(2, int(400 / (N * K * 2 / 1e6)) + 1)
    for rep in range(2):
        print(f"{N:5d} x {K:4d}   SwiGLU strip pairs, {N // 32:4d} workgroups   {run(N, K, 2, 1, 1, 16):12.2f} {run(N, K, 2, 1, n, 16):13.2f}", flush=True)
        print(f"{N:5d} x {K:4d}   plain 16-feature strips, {N // 16:4d} wgs     {run(N, K, 0, 1, 1, 16):12.2f} {run(N, K, 0, 1, n, 16):13.2f}", flush=True)
        print(f"{N:5d} x {K:4d}   plain  8-feature strips, {N // 8:4d} wgs     {run(N, K, 0, 1, 1, 8):12.2f} {run(N, K, 0, 1, n, 8):13.2f}", flush=True)
EOF
echo "rc=$?"; grep -v amdgpu.ids "$OUT/gu_wgs.log"
