#!/usr/bin/env python3
"""Diagnostic for the contention test (tests/test_gpu_parity.py): which engine configuration stops being run-to-run identical when other
work shares the device, and where the codes first differ.  0.6B dims, batch 8, 40 frames teacher-forced.

    python tools/diag_contention.py
"""
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
gc = torch.from_numpy(g["codes"][:, :40].copy())
sup = [i for i in range(cfg.vocab_size - 1024, cfg.vocab_size) if i != cfg.codec_eos_token_id]


def mk(**opts):
    with _lib.options(**opts):
        return TalkerEngine(cfg, wn, weight_dtype=torch.bfloat16, device=dev, max_batch=len(lens), max_seq=256, use_graph=True)


def run(e, teacher=True):
    if teacher:
        return e.generate(emb, mask, tr, pad, teacher_codes=gc, suppress_tokens=sup).own.cpu().numpy()
    return e.generate(emb, mask, tr, pad, max_new_tokens=41, min_new_tokens=41, do_sample=False, subtalker_dosample=False,
                      suppress_tokens=sup).codes.cpu().numpy()


def first_diff(a, b):
    d = np.argwhere(a != b)
    if len(d) == 0:
        return None
    fr = int(d[:, 1].min())
    rows = sorted(set(int(x[0]) for x in d if x[1] == fr))
    cbs = sorted(set(int(x[2]) for x in d if x[1] == fr))
    return {"first_frame": fr, "rows": rows, "codebooks": cbs, "differing": int(len(d)), "of": int(a.size)}


def background(fn):
    stop = threading.Event()
    n = [0]
    def loop():
        while not stop.is_set():
            fn(); n[0] += 1
    t = threading.Thread(target=loop); t.start()
    return stop, t, n


def check(name, e, others, teacher=True, reps=4):
    ref = run(e, teacher)
    bgs = [background(o) for o in others]
    time.sleep(0.2)
    outs = [run(e, teacher) for _ in range(reps)]
    for stop, t, n in bgs:
        stop.set()
    for stop, t, n in bgs:
        t.join()
    diffs = [first_diff(o, ref) for o in outs]
    print(f"[diag] {name:60s} background loops {[n[0] for _, _, n in bgs]}  diffs vs quiet run: {diffs}", flush=True)


sep = mk(QTTS_CP_ATTN_O="0", QTTS_CP_MLP="0")
fused = mk(QTTS_CP_MLP="0")
sep2 = mk(QTTS_CP_ATTN_O="0", QTTS_CP_MLP="0")
fused2 = mk(QTTS_CP_MLP="0")
print("[diag] stats:", {k: v for k, v in sep.stats().items() if k.startswith("cp_")}, {k: v for k, v in fused.stats().items() if k.startswith("cp_")}, flush=True)
ccfg = synth.codec_real()
codec = CodecDecoderEngine(ccfg, {k: torch.from_numpy(v) for k, v in synth.codec_weights(ccfg).items()}, compute_dtype=torch.bfloat16, device=dev, max_batch=8, max_frames=150)
codes = torch.from_numpy(np.random.default_rng(3).integers(0, ccfg.codebook_size, (8, ccfg.num_quantizers, 125))).to(dev)
cstream = torch.cuda.Stream(device=dev)
def codec_fn():
    with torch.cuda.stream(cstream):
        codec.forward(codes); cstream.synchronize()

for teacher in (True, False):
    tag = "teacher-forced (eager)" if teacher else "free greedy (graph)"
    check(f"{tag}: separate launches, alone", sep, [], teacher)
    check(f"{tag}: fused, alone", fused, [], teacher)
    check(f"{tag}: separate + codec loop", sep, [codec_fn], teacher)
    check(f"{tag}: separate + other separate engine", sep, [lambda: run(sep2, teacher)], teacher)
    check(f"{tag}: separate + fused engine", sep, [lambda: run(fused2, teacher)], teacher)
    check(f"{tag}: fused + separate engine", fused, [lambda: run(sep2, teacher)], teacher)
    check(f"{tag}: fused + fused engine", fused, [lambda: run(fused2, teacher)], teacher)
    check(f"{tag}: fused + codec + fused + separate", fused, [codec_fn, lambda: run(fused2, teacher), lambda: run(sep2, teacher)], teacher)
    check(f"{tag}: separate + codec + fused + separate", sep, [codec_fn, lambda: run(fused2, teacher), lambda: run(sep2, teacher)], teacher)
for e in (sep, fused, sep2, fused2):
    print("[diag] giveups", e.stats()["cp_fused_giveups"])
