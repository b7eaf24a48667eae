#!/usr/bin/env python3
"""Bisect the separate-launch path under a concurrent codec decode: which of its kernels makes the codes run-to-run different?"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import synth
from qwen3_tts_amd import _lib
from qwen3_tts_amd.talker import TalkerEngine
from qwen3_tts_amd.codec import CodecDecoderEngine

dev = "cuda:0"
cfg = synth.talker_06b()
g = np.load(os.path.join(ROOT, "tests", "golden", "talker_06b_b8.npz"))
wn = {k: torch.from_numpy(v) for k, v in synth.talker_weights(cfg, with_text=False).items()}
lens = [int(x) for x in g["lens"]]
emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(int(g["seed"])), cfg, lens, int(g["n_trail"]), scale=0.05)
sup = [i for i in range(cfg.vocab_size - 1024, cfg.vocab_size) if i != cfg.codec_eos_token_id]
ccfg = synth.codec_real()
cw = {k: torch.from_numpy(v) for k, v in synth.codec_weights(ccfg).items()}
cstream = torch.cuda.Stream(device=dev)
codec = CodecDecoderEngine(ccfg, cw, compute_dtype=torch.bfloat16, device=dev, max_batch=8, max_frames=150)
codes8s = torch.from_numpy(np.random.default_rng(3).integers(0, ccfg.codebook_size, (8, ccfg.num_quantizers, 12))).to(dev)


def run(e):
    return e.generate(emb, mask, tr, pad, max_new_tokens=41, min_new_tokens=41, do_sample=False, subtalker_dosample=False, suppress_tokens=sup).codes.cpu().numpy()


def under(fn, reps):
    stop = threading.Event(); n = [0]
    def loop():
        with torch.cuda.stream(cstream):
            while not stop.is_set():
                codec.forward(codes8s); cstream.synchronize(); n[0] += 1
    t = threading.Thread(target=loop); t.start(); time.sleep(0.2)
    try:
        return [fn() for _ in range(reps)], n[0]
    finally:
        stop.set(); t.join()


def ndiff(a, b):
    d = np.argwhere(a != b)
    return None if len(d) == 0 else (int(d[:, 1].min()), int(len(d)))


SEP = dict(QTTS_CP_ATTN_O="0", QTTS_CP_MLP="0")
CASES = [("separate launches (attn_cp)", SEP), ("separate launches (attn_cp), again", SEP), ("fused (default)", dict())]
for name, opts in CASES:
    for k, v in opts.items():
        _lib.set_option(k, v)
    codec = CodecDecoderEngine(ccfg, cw, compute_dtype=torch.bfloat16, device=dev, max_batch=8, max_frames=150)
    e = TalkerEngine(cfg, wn, weight_dtype=torch.bfloat16, device=dev, max_batch=len(lens), max_seq=256, use_graph="eager" not in name)
    quiet = run(e)
    outs, n = under(lambda: run(e), 16)
    for k in opts:
        _lib.set_option(k, None)
    print("   codec stats:", codec.stats(), flush=True)
    bad = sum(o is not None for o in [ndiff(o, quiet) for o in outs])
    print(f"[diag4] {name:64s} codec loops {n:4d}; runs that differ from the quiet run: {bad} of 16  {[ndiff(o, quiet) for o in outs]}", flush=True)
    del e
    torch.cuda.empty_cache()
