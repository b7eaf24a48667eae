#!/usr/bin/env python3
"""Which kernel of the SEPARATE-launch code-predictor path stops being run-to-run identical while a codec decode loops on another stream
(tools/diag_contention.py found: only that path, only with the codec beside it).  One engine per switch setting, free greedy generation
through the captured frame graph, 6 runs against the quiet run."""
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
codes = torch.from_numpy(np.random.default_rng(3).integers(0, ccfg.codebook_size, (8, ccfg.num_quantizers, 125))).to(dev)
cstream = torch.cuda.Stream(device=dev)


def run(e):
    o = e.generate(emb, mask, tr, pad, max_new_tokens=41, min_new_tokens=41, do_sample=False, subtalker_dosample=False, suppress_tokens=sup,
                   output_hidden_states=True)
    return o.codes.cpu().numpy(), o.hidden.cpu().numpy()


def with_codec(codec, fn, reps):
    stop = threading.Event(); n = [0]
    def loop():
        with torch.cuda.stream(cstream):
            while not stop.is_set():
                codec.forward(codes); cstream.synchronize(); n[0] += 1
    t = threading.Thread(target=loop); t.start(); time.sleep(0.2)
    try:
        return [fn() for _ in range(reps)], n[0]
    finally:
        stop.set(); t.join()


def first_diff(a, b):
    d = np.argwhere(a != b)
    if len(d) == 0:
        return None
    fr = int(d[:, 1].min())
    return (fr, sorted(set(int(x[0]) for x in d if x[1] == fr)), sorted(set(int(x[2]) for x in d if x[1] == fr))[:4])


CASES = [("separate (baseline)", dict(QTTS_CP_ATTN_O="0", QTTS_CP_MLP="0")),
         ("fused attn+o WITHOUT the q|k|v front (decode GEMM q|k|v)", dict(QTTS_CP_ATTN_O="1", QTTS_CP_FRONT="0", QTTS_CP_MLP="0")),
         ("separate, generic decode GEMM (QTTS_SKINNY8=0)", dict(QTTS_CP_ATTN_O="0", QTTS_CP_MLP="0", QTTS_SKINNY8="0")),
         ("separate, strip floor 48 (o-projection in 16-feature strips)", dict(QTTS_CP_ATTN_O="0", QTTS_CP_MLP="0", QTTS_FS_MIN_WGS="48")),
         ("separate, no rope table / kernarg etc. unchanged, codec fp32", dict(QTTS_CP_ATTN_O="0", QTTS_CP_MLP="0")),
         ("fused (default incl. MLP)", dict())]
for k, (name, opts) in enumerate(CASES):
    cdt = torch.float32 if "codec fp32" in name else torch.bfloat16
    codec = CodecDecoderEngine(ccfg, cw, compute_dtype=cdt, device=dev, max_batch=8, max_frames=150)
    for key, v in opts.items():
        _lib.set_option(key, v)
    e = TalkerEngine(cfg, wn, weight_dtype=torch.bfloat16, device=dev, max_batch=len(lens), max_seq=256, use_graph=True)
    quiet_c, quiet_h = run(e)
    again_c, _ = run(e)
    outs, nloops = with_codec(codec, lambda: run(e), 6)
    for key in opts:
        _lib.set_option(key, None)
    diffs = [first_diff(c, quiet_c) for c, _ in outs]
    hd = [float(np.abs(h[:, :3] - quiet_h[:, :3]).max()) for _, h in outs]
    print(f"[diag2] {name:62s} quiet repeat identical {np.array_equal(again_c, quiet_c)}; codec loops {nloops}; first diffs (frame, rows, codebooks) {diffs}; "
          f"max |hidden diff| in frames 0-2: {hd}", flush=True)
    del e, codec
    torch.cuda.empty_cache()
