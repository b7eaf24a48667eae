"""Perf iteration tool (not a test, not the bench): build the talker/codec at real dims with cheap tiled-random
weights and time the AR loop and the codec decode.  `--prof` prints the per-launch HIP-event average of the
skinny GEMM.  Usage on the GPU box:  python tools/perf_frame.py --model 1.7b --frames 40 [--codec]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import synth
from qwen3_tts_amd.talker import TalkerEngine
from qwen3_tts_amd.codec import CodecDecoderEngine

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="1.7b"); ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--codec", action="store_true")
ap.add_argument("--talker", action="store_true"); ap.add_argument("--prof", action="store_true")
ap.add_argument("--no-graph", action="store_true"); ap.add_argument("--codec-dtype", default="bf16")
ap.add_argument("--codec-frames", type=int, default=125); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--talker-dtype", default="bf16", choices=["bf16", "f32"], help="f32 = the exact-fp32 parity mode")
a = ap.parse_args()
if not (a.codec or a.talker): a.talker = True

def cheap(shapes, std_of):
    base = np.random.default_rng(0).standard_normal(1 << 20, dtype=np.float32)
    out = {}
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        v = np.resize(base, n).reshape(shp) * np.float32(std_of(k, shp))
        if "norm" in k and k.endswith("weight"): v = v * 0 + 1
        if "head" in k:      # the tiled base repeats every 2^20 values: identical head rows = tied logits, which the sampler sees
            v = np.random.default_rng(__import__("zlib").crc32(k.encode())).standard_normal(shp, dtype=np.float32) * np.float32(std_of(k, shp))
        if k.endswith("cluster_usage"): v = np.abs(v) + 0.5
        out[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out

B, F = a.batch, a.frames
if a.talker:
    t = {"1.7b": synth.talker_17b, "0.6b": synth.talker_06b}[a.model]()
    t0 = time.time()
    w = cheap(synth.talker_param_shapes(t, with_text=False), lambda k, s: 0.08 if ("head" in k) else 0.02)
    eng = TalkerEngine(t, w, weight_dtype=torch.bfloat16 if a.talker_dtype == "bf16" else torch.float32, max_batch=B, max_seq=64 + F + 8, use_graph=not a.no_graph)
    del w
    print(f"talker build {time.time() - t0:.1f}s", flush=True)
    lens = [24 + 4 * (i % 8) + 12 for i in range(B)]
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(1), t, lens, 1)
    sup = [i for i in range(t.vocab_size - 1024, t.vocab_size) if i != t.codec_eos_token_id]
    kw = dict(max_new_tokens=F + 1, min_new_tokens=F + 1, suppress_tokens=sup, output_hidden_states=False)
    for mode, extra in (("sampling", {}), ("greedy", dict(do_sample=False, subtalker_dosample=False))):
        eng.generate(emb, mask, tr, pad, seed=0, **kw, **extra); torch.cuda.synchronize()
        ts = []
        for r in range(a.reps):
            t1 = time.perf_counter(); o = eng.generate(emb, mask, tr, pad, seed=r, **kw, **extra); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        st = eng.stats(); dt = min(ts)
        # subtract prefill by timing a 1-token call
        t1 = time.perf_counter(); eng.generate(emb, mask, tr, pad, seed=0, **dict(kw, max_new_tokens=1, min_new_tokens=1), **extra); torch.cuda.synchronize()
        tp = time.perf_counter() - t1
        ms = 1000 * (dt - tp) / F
        print(f"[{mode}] total {1000*dt:.1f} ms, prefill+1tok {1000*tp:.1f} ms, {ms:.3f} ms/frame, graph_nodes {st['graph_nodes']}, "
              f"weights {st['weight_bytes_per_frame']/1e9:.2f} GB/frame -> {st['weight_bytes_per_frame']/ms/1e6:.0f} GB/s, "
              f"{B*16/ms*1000:.0f} tok/s, RTFx {B*0.08/ms*1000:.0f}", flush=True)
    if a.prof:
        eng.set_profile(1)
        eng.generate(emb, mask, tr, pad, seed=0, **dict(kw, max_new_tokens=9, min_new_tokens=9)); eng.set_profile(0)
        cls = eng.gemm_profile()
        tot = sum(c["total_ms"] for c in cls); n = sum(c["launches"] for c in cls); b = sum(c["launches"] * c["bytes_per_launch"] for c in cls)
        for c in sorted(cls, key=lambda c: (c["stack"], -c["bytes_per_launch"])):
            us = 1e3 * c["total_ms"] / c["launches"]
            print(f"  stack {c['stack']} N={c['N']:5d} K={c['K']:5d} x{c['launches'] // 6:3d}/frame  {us:6.2f} us  {c['bytes_per_launch'] / us / 1e3:6.0f} GB/s")
        print(f"decode GEMM in the real frame step: {n // 6} launches/frame, avg {1e3 * tot / n:.2f} us/launch, {b / tot / 1e6:.0f} GB/s")
if a.codec:
    c = synth.codec_real()
    t0 = time.time()
    def cstd(k, s):
        if k.endswith(".weight") and len(s) >= 2: return 1.0 / np.sqrt(np.prod(s[1:]) if ".block.1." not in k else s[0] * 2) * (0.35 if "conv2" in k else 1.0)
        if k.endswith("alpha") or k.endswith("beta"): return 0.3
        return 0.05
    w = cheap(synth.codec_param_shapes(c), cstd)
    Tc = a.codec_frames
    dec = CodecDecoderEngine(c, w, compute_dtype=torch.bfloat16 if a.codec_dtype == "bf16" else torch.float32, max_batch=B, max_frames=min(Tc, 300) + 25)
    print(f"codec build {time.time() - t0:.1f}s", flush=True)
    codes = torch.randint(0, 2048, (B, Tc, 16)).cuda()
    dec.decode_padded(codes); torch.cuda.synchronize()
    ts = []
    for r in range(a.reps):
        t1 = time.perf_counter(); wav, _ = dec.decode_padded(codes); torch.cuda.synchronize(); ts.append(time.perf_counter() - t1)
    dt = min(ts); gf = 5.12 * Tc * B
    print(f"[codec {a.codec_dtype}] B={B} T={Tc}: {1000*dt:.2f} ms -> {gf/dt/1000:.1f} TFLOP/s algorithmic, RTFx {B*Tc*0.08/dt:.0f}, finite={bool(torch.isfinite(wav).all())}")
