#!/usr/bin/env python3
"""Which XCD runs workgroup i of a launch (HW_REG_XCC_ID)?  cp_mlp.hip slices its exchange by blockIdx % 8; quiet, and with a codec decode
looping on another stream."""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import synth
from qwen3_tts_amd import _lib
from qwen3_tts_amd.codec import CodecDecoderEngine
lib = _lib.load_library()
lib.qtts_debug_xcc_map.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.c_void_p]
torch.cuda.set_device(0)
st = torch.cuda.Stream()

def probe(grid):
    out = (C.c_int32 * grid)()
    _lib.check(lib.qtts_debug_xcc_map(grid, out, C.c_void_p(st.cuda_stream)))
    return np.array(out[:])

for grid in (64, 256, 768):
    maps = [probe(grid) for _ in range(20)]
    ok = [bool((m == np.arange(grid) % 8).all()) for m in maps]
    rot = [int(m[0]) for m in maps]
    cyc = [bool((((m - m[0]) % 8) == np.arange(grid) % 8).all()) for m in maps]
    print(f"[xcc] quiet, grid {grid}: blockIdx % 8 == XCC_ID in {sum(ok)}/20 launches; (XCC_ID - XCC_ID[0]) % 8 == blockIdx % 8 in {sum(cyc)}/20; XCC_ID of workgroup 0: {rot}")
ccfg = synth.codec_real()
codec = CodecDecoderEngine(ccfg, {k: torch.from_numpy(v) for k, v in synth.codec_weights(ccfg).items()}, compute_dtype=torch.bfloat16, device="cuda:0", max_batch=8, max_frames=150)
codes = torch.from_numpy(np.random.default_rng(3).integers(0, ccfg.codebook_size, (8, ccfg.num_quantizers, 12))).cuda()
cs = torch.cuda.Stream()
stop = threading.Event()
def loop():
    with torch.cuda.stream(cs):
        while not stop.is_set():
            codec.forward(codes); cs.synchronize()
t = threading.Thread(target=loop); t.start(); time.sleep(0.2)
try:
    for grid in (256,):
        maps = [probe(grid) for _ in range(200)]
        ok = sum(bool((m == np.arange(grid) % 8).all()) for m in maps)
        cyc = sum(bool((((m - m[0]) % 8) == np.arange(grid) % 8).all()) for m in maps)
        print(f"[xcc] codec neighbour, grid {grid}: blockIdx % 8 == XCC_ID in {ok}/200 launches; round-robin from another start in {cyc}/200; starts seen {sorted(set(int(m[0]) for m in maps))}")
finally:
    stop.set(); t.join()
