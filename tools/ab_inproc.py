#!/usr/bin/env python3
"""In-process, interleaved A/B of engine-level knobs (process-to-process variance on the pool's boxes is ~5 %, larger than the
effects being measured): for each repetition, for each configuration: set the environment, build a 1.7B bf16 talker engine
(the knobs are read at engine construction), time `--frames` frame steps (min of 3), destroy the engine.

    python tools/ab_inproc.py --frames 40 --reps 3
"""
import os as _os
_os.environ.setdefault("QTTS_DEBUG_ENV_LIVE", "1")   # the library copies its A/B switches once unless told otherwise (csrc/common.h QTTS_ENV)
import argparse, gc, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import synth
from qwen3_tts_amd.talker import TalkerEngine

CONFIGS = {
    "default": {},                               # 4 waves per workgroup below 16 MB, 8 above
    "skinny8_nw8": {"QTTS_SKINNY8_NW": "8"},
    "skinny8_nw4": {"QTTS_SKINNY8_NW": "4"},
}
KEYS = sorted({k for c in CONFIGS.values() for k in c})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40); ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    t = synth.talker_17b()
    base = np.random.default_rng(0).standard_normal(1 << 20, dtype=np.float32)
    w = {}
    for k, shp in synth.talker_param_shapes(t, with_text=False).items():
        v = np.resize(base, int(np.prod(shp))).reshape(shp) * np.float32(0.08 if "head" in k else 0.02)
        if "norm" in k and k.endswith("weight"): v = v * 0 + 1
        if "head" in k: v = np.random.default_rng(__import__("zlib").crc32(k.encode())).standard_normal(shp, dtype=np.float32) * np.float32(0.08)   # no tied logits
        w[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    B, F = 8, a.frames
    lens = [24 + 4 * (i % 8) + 12 for i in range(B)]
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(1), t, lens, 1)
    sup = [i for i in range(t.vocab_size - 1024, t.vocab_size) if i != t.codec_eos_token_id]
    kw = dict(max_new_tokens=F + 1, min_new_tokens=F + 1, suppress_tokens=sup, output_hidden_states=False)
    names = [n for n in CONFIGS if not a.only or n in a.only]
    res = {n: [] for n in names}
    for rep in range(a.reps):
        for n in names:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(CONFIGS[n])
            eng = TalkerEngine(t, w, weight_dtype=torch.bfloat16, max_batch=B, max_seq=64 + F + 8, use_graph=True)
            eng.generate(emb, mask, tr, pad, seed=0, **kw); torch.cuda.synchronize()
            ts = []
            for r in range(3):
                t1 = time.perf_counter(); eng.generate(emb, mask, tr, pad, seed=r, **kw); torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            t1 = time.perf_counter(); eng.generate(emb, mask, tr, pad, seed=0, **dict(kw, max_new_tokens=1, min_new_tokens=1)); torch.cuda.synchronize()
            tp = time.perf_counter() - t1
            ms = 1000 * (min(ts) - tp) / F
            res[n].append(round(ms, 4))
            print(f"[ab_inproc] rep {rep} {n:28s} {ms:.3f} ms/frame", flush=True)
            del eng; gc.collect(); torch.cuda.empty_cache()
    out = {n: {"ms_per_frame": v, "min": min(v), "median": float(np.median(v))} for n, v in res.items()}
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ab_inproc.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
