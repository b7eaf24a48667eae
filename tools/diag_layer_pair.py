#!/usr/bin/env python3
"""Diagnostic (round 6): under which neighbours does the layer launch (csrc/cp_layer.hip) give up?  0.6B dims, batch 8, 40 free-running greedy
frames through the captured graph; for each case fresh engines (a give-up retires an engine), three generations of the engine under
test while the neighbours loop, then the engines' give-up counters and whether the codes equalled the quiet run.

    python tools/diag_layer_pair.py
"""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import synth
from qwen3_tts_amd import _lib
from qwen3_tts_amd.talker import TalkerEngine
from qwen3_tts_amd.codec import CodecDecoderEngine

dev = "cuda:0"
cfg = synth.talker_06b()
wn = {k: torch.from_numpy(v) for k, v in synth.talker_weights(cfg, with_text=False).items()}
lens = [30, 41, 25, 37, 44, 28, 33, 39]
emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(5), cfg, lens, 2, scale=0.05)
sup = [i for i in range(cfg.vocab_size - 1024, cfg.vocab_size) if i != cfg.codec_eos_token_id]
lib = _lib.load_library()
lib.qtts_debug_cp_layer_occupancy.argtypes = [C.c_int32, C.c_int32, C.c_int32]
print("[diag] occupancy API, workgroups of the layer launch per compute unit (bf16, fp32):",
      lib.qtts_debug_cp_layer_occupancy(cfg.cp_hidden_size, cfg.cp_intermediate_size, 1), lib.qtts_debug_cp_layer_occupancy(cfg.cp_hidden_size, cfg.cp_intermediate_size, 0), flush=True)


def mk(**opts):
    with _lib.options(**opts):
        return TalkerEngine(cfg, wn, weight_dtype=torch.bfloat16, device=dev, max_batch=len(lens), max_seq=256, use_graph=True)


def run(e):
    return e.generate(emb, mask, tr, pad, max_new_tokens=41, min_new_tokens=41, do_sample=False, subtalker_dosample=False, suppress_tokens=sup).codes.cpu().numpy()


ccfg = synth.codec_real()
codec = CodecDecoderEngine(ccfg, {k: torch.from_numpy(v) for k, v in synth.codec_weights(ccfg).items()}, compute_dtype=torch.bfloat16, device=dev, max_batch=8, max_frames=150)
ccodes = {n: torch.from_numpy(np.random.default_rng(3).integers(0, ccfg.codebook_size, (8, ccfg.num_quantizers, n))).to(dev) for n in (12, 125)}
cstream = torch.cuda.Stream(device=dev)


def codec_fn(n):
    def f():
        with torch.cuda.stream(cstream):
            codec.forward(ccodes[n]); cstream.synchronize()
    return f


KIND = {"layer": {}, "two": {"QTTS_CP_LAYER": "0"}, "sep": {"QTTS_CP_ATTN_O": "0", "QTTS_CP_MLP": "0"}}


def case(name, under_test, neighbours):
    e = mk(**KIND[under_test])
    quiet = run(e)
    others = []
    for n in neighbours:
        if n.startswith("codec"):
            others.append((n, None, codec_fn(int(n[5:]))))
        else:
            o = mk(**KIND[n])
            run(o)
            others.append((n, o, (lambda oo: (lambda: run(oo)))(o)))
    stop = threading.Event()
    loops = [0] * len(others)
    errs = []

    def loop(i, fn):
        try:
            while not stop.is_set():
                fn(); loops[i] += 1
        except Exception as ex:        # noqa: BLE001
            errs.append(repr(ex)[:120])
    ths = [threading.Thread(target=loop, args=(i, fn)) for i, (_, _, fn) in enumerate(others)]
    for t in ths: t.start()
    time.sleep(0.1)
    t0 = time.perf_counter()
    same = []
    for _ in range(3):
        same.append(bool(np.array_equal(run(e), quiet)))
    dt = (time.perf_counter() - t0) / 3
    stop.set()
    for t in ths: t.join()
    st = e.stats()
    print(f"[diag] {name:46s} under test: layer/step {st['cp_layer_per_step']:2d} fused/step {st['cp_fused_per_step']:2d} give-ups {st['cp_fused_giveups']}  == quiet {same}  "
          f"{1e3 * dt:7.1f} ms/generation  neighbours: " + ", ".join(f"{n} loops {loops[i]} give-ups {o.stats()['cp_fused_giveups'] if o is not None else '-'}" for i, (n, o, _) in enumerate(others))
          + (f"  errors {errs}" if errs else ""), flush=True)
    for _, o, _ in others:
        del o
    del e
    torch.cuda.empty_cache()


case("layer alone", "layer", [])
case("layer + codec 8 x 12", "layer", ["codec12"])
case("layer + codec 8 x 125", "layer", ["codec125"])
case("layer + separate-launch engine", "layer", ["sep"])
case("layer + two-launch fused engine", "layer", ["two"])
case("layer + layer", "layer", ["layer"])
case("two-launch + two-launch (round 5)", "two", ["two"])
case("layer + layer + codec 8 x 12 + separate", "layer", ["layer", "codec12", "sep"])
case("layer + two-launch + codec 8 x 12 + separate", "layer", ["two", "codec12", "sep"])
