"""In-kernel timestamps of the codec decoder's tap-reuse GEMM (`gemm_tap2_kernel`) on the `tstamp` build variant: per layer class
(K x taps, N) the prologue, the k-loop (per step) and the epilogue, for the first and the last workgroup of every launch.

    python qwen3-tts_amd/build.py --variant tstamp && python tools/ts_codec.py [--batch 8] [--frames 125]
"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "qwen3-tts_amd", "libqtts_tstamp.so")
os.environ["QTTS_LIBRARY"] = LIB
import numpy as np, torch
import synth
from qwen3_tts_amd.codec import CodecDecoderEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--frames", type=int, default=125); ap.add_argument("--json", default=None)
a = ap.parse_args()
REC = np.dtype([("t", "<u8", 6), ("kind", "<i4"), ("a", "<i4"), ("b", "<i4"), ("blk", "<i4")])
lib = C.CDLL(LIB)
fn = lib.qtts_debug_tslog_tap; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int


def drain():
    buf = np.zeros(1 << 15, dtype=REC)
    n = fn(buf.ctypes.data, 1 << 15)
    if n < 0: raise RuntimeError("qtts_debug_tslog_tap failed")
    return buf[:n].copy()


c = synth.codec_real()
base = np.random.default_rng(0).standard_normal(1 << 20, dtype=np.float32)
def cstd(k, s):
    if k.endswith(".weight") and len(s) >= 2: return 1.0 / np.sqrt(np.prod(s[1:]) if ".block.1." not in k else s[0] * 2) * (0.35 if "conv2" in k else 1.0)
    if k.endswith("alpha") or k.endswith("beta"): return 0.3
    return 0.05
w = {}
for k, shp in synth.codec_param_shapes(c).items():
    v = np.resize(base, int(np.prod(shp))).reshape(shp) * np.float32(cstd(k, shp))
    if "norm" in k and k.endswith("weight"): v = v * 0 + 1
    if k.endswith("cluster_usage"): v = np.abs(v) + 0.5
    w[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
dec = CodecDecoderEngine(c, w, compute_dtype=torch.bfloat16, max_batch=a.batch, max_frames=min(a.frames, 300) + 25)
codes = torch.randint(0, 2048, (a.batch, a.frames, 16)).cuda()
dec.decode_padded(codes); torch.cuda.synchronize(); drain()
dec.decode_padded(codes); torch.cuda.synchronize()
r = drain()
print(f"{len(r)} records (first / last workgroup of {len(r) // 2} gemm_tap2 launches), batch {a.batch}, {a.frames} frames")
US = 0.01
rows = []
print(f"{'K x taps':>9s} {'N':>5s} {'n':>4s} {'steps':>6s} {'prologue':>9s} {'us/step':>8s} {'k-loop':>8s} {'epilogue':>9s} {'in-kernel':>10s}   (MFMA time of a step: 16 x 16x16x32 per wave ~ 0.21 us)")
for key in sorted({(int(x), int(y)) for x, y in zip(r["a"], r["b"])}):
    rr = r[(r["a"] == key[0]) & (r["b"] == key[1])]
    t = rr["t"].astype(np.float64)
    steps = key[0] // 32
    pro = (t[:, 1] - t[:, 0]).mean() * US
    loop = (t[:, 4] - t[:, 1]).mean() * US
    epi = (t[:, 5] - t[:, 4]).mean() * US
    tot = (t[:, 5] - t[:, 0]).mean() * US
    rows.append(dict(k_taps=key[0], n=key[1], records=len(rr), steps=steps, prologue_us=round(pro, 2), us_per_step=round(loop / steps, 3),
                     loop_us=round(loop, 1), epilogue_us=round(epi, 1), in_kernel_us=round(tot, 1)))
    print(f"{key[0]:9d} {key[1]:5d} {len(rr):4d} {steps:6d} {pro:9.2f} {loop / steps:8.3f} {loop:8.1f} {epi:9.1f} {tot:10.1f}")
if a.json: json.dump(dict(batch=a.batch, frames=a.frames, rows=rows), open(a.json, "w"), indent=1)

# ---- the fused residual unit (resunit.hip): a sample of workgroups of every launch
fr = lib.qtts_debug_tslog_ru; fr.argtypes = [C.c_void_p, C.c_int]; fr.restype = C.c_int
buf = np.zeros(1 << 15, dtype=REC)
n = fr(buf.ctypes.data, 1 << 15)
r = buf[:max(n, 0)]
print(f"\nresunit_kernel: {len(r)} sampled workgroups (both decodes)")
print(f"{'NC,TM':>6s} {'dil':>4s} {'n':>4s} {'staged':>8s} {'step 0':>8s} {'conv7 (rest)':>13s} {'1x1':>7s} {'epilogue':>9s} {'in-kernel':>10s}  us")
ru_rows = []
for key in sorted({(int(x), int(y)) for x, y in zip(r["a"], r["b"])}):
    rr = r[(r["a"] == key[0]) & (r["b"] == key[1])]
    t = rr["t"].astype(np.float64) * US
    d = [(t[:, i + 1] - t[:, i]).mean() for i in range(5)]
    ru_rows.append(dict(nc_tm=key[0], dil=key[1], records=len(rr), staged_us=round(d[0], 2), step0_us=round(d[1], 2), conv7_rest_us=round(d[2], 2),
                        conv1x1_us=round(d[3], 2), epilogue_us=round(d[4], 2), in_kernel_us=round(sum(d), 2)))
    print(f"{key[0]:6d} {key[1]:4d} {len(rr):4d} {d[0]:8.2f} {d[1]:8.2f} {d[2]:13.2f} {d[3]:7.2f} {d[4]:9.2f} {sum(d):10.2f}")
if a.json: json.dump(dict(batch=a.batch, frames=a.frames, rows=rows, resunit=ru_rows), open(a.json, "w"), indent=1)
