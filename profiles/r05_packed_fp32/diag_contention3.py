#!/usr/bin/env python3
"""Is the separate-launch path's run-to-run difference under a concurrent codec decode a matter of TIMING (any busy neighbour triggers it)
or of the codec's own kernels (something the codec writes)?  Separate-launch engine, free greedy through the frame graph, 8 runs against
the quiet run under different neighbours on another stream."""
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


def run(e):
    return e.generate(emb, mask, tr, pad, max_new_tokens=41, min_new_tokens=41, do_sample=False, subtalker_dosample=False, suppress_tokens=sup).codes.cpu().numpy()


def under(bg, fn, reps):
    stop = threading.Event(); n = [0]
    def loop():
        with torch.cuda.stream(cstream):
            while not stop.is_set():
                bg(); cstream.synchronize(); n[0] += 1
    t = threading.Thread(target=loop); t.start(); time.sleep(0.2)
    try:
        return [fn() for _ in range(reps)], n[0]
    finally:
        stop.set(); t.join()


def ndiff(a, b):
    d = np.argwhere(a != b)
    return None if len(d) == 0 else (int(d[:, 1].min()), int(len(d)))


with _lib.options(QTTS_CP_ATTN_O="0", QTTS_CP_MLP="0"):
    e = TalkerEngine(cfg, wn, weight_dtype=torch.bfloat16, device=dev, max_batch=len(lens), max_seq=256, use_graph=True)
quiet = run(e)
A = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16); Bm = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
big = torch.empty(1 << 28, device=dev, dtype=torch.uint8)
codec = CodecDecoderEngine(ccfg, cw, compute_dtype=torch.bfloat16, device=dev, max_batch=8, max_frames=150)
codes8 = torch.from_numpy(np.random.default_rng(3).integers(0, ccfg.codebook_size, (8, ccfg.num_quantizers, 125))).to(dev)
codes1 = codes8[:1].contiguous()
codes8s = codes8[:, :, :12].contiguous()
NEIGH = [("torch bf16 matmul 4096^3", lambda: torch.matmul(A, Bm)),
         ("torch memset 256 MB", lambda: big.fill_(1)),
         ("codec bf16 8 x 125 frames", lambda: codec.forward(codes8)),
         ("codec bf16 1 x 125 frames", lambda: codec.forward(codes1)),
         ("codec bf16 8 x 12 frames", lambda: codec.forward(codes8s)),
         ("codec bf16 8 x 125 frames, no codec graph", None)]
for name, bg in NEIGH:
    if bg is None:
        _lib.set_option("QTTS_CODEC_GRAPH", "0")
        bg = lambda: codec.forward(codes8)
    outs, n = under(bg, lambda: run(e), 8)
    print(f"[diag3] neighbour: {name:44s} loops {n:5d}  (first differing frame, differing codes) per run: {[ndiff(o, quiet) for o in outs]}", flush=True)
_lib.set_option("QTTS_CODEC_GRAPH", None)
print("[diag3] quiet again:", ndiff(run(e), quiet))
