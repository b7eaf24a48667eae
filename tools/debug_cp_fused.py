"""Debug helper (round 4): first frames of a free-running greedy bf16 generation at 0.6B dims with the fused code-predictor launch
on / front off / off, for several batch sizes, eager and graph.  Prints the agreement of every variant with the separate launches."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from qwen3_tts_amd.talker import TalkerEngine
from qwen3_tts_amd import _lib

cfg = synth.talker_06b()
wn = {k: torch.from_numpy(v) for k, v in synth.talker_weights(cfg, with_text=False).items()}
sup = [i for i in range(cfg.vocab_size - 1024, cfg.vocab_size) if i != cfg.codec_eos_token_id]
dev = torch.device("cuda:0")
for B in (3,):
    lens = [36 + 4 * i for i in range(B)]
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(5), cfg, lens, 2, scale=0.05)
    res = {}
    for name, env in (("plain", {"QTTS_CP_ATTN_O": "0"}), ("attn_o", {"QTTS_CP_FRONT": "0"}), ("front", {})):
        for graph in (False, True):
            for k in ("QTTS_CP_ATTN_O", "QTTS_CP_FRONT"): _lib.set_option(k, None)
            for k, v in env.items(): _lib.set_option(k, v)
            eng = TalkerEngine(cfg, wn, weight_dtype=torch.bfloat16, device=dev, max_batch=B, max_seq=256, use_graph=graph)
            out = eng.generate(emb, mask, tr, pad, max_new_tokens=7, min_new_tokens=7, do_sample=False, subtalker_dosample=False, suppress_tokens=sup)
            res[(name, graph)] = out.codes.cpu().numpy()
            del eng; torch.cuda.empty_cache()
    ref = res[("plain", True)]
    print(f"B={B}: plain eager==graph {np.array_equal(res[('plain', False)], ref)}")
    for name in ("attn_o", "front"):
        for graph in (False, True):
            c = res[(name, graph)]
            n = min(c.shape[1], ref.shape[1])
            per_frame = [(c[:, f] == ref[:, f]).mean() for f in range(n)]
            print(f"  {name:6s} graph={int(graph)}: agreement per frame " + " ".join(f"{x:.2f}" for x in per_frame) + f"   frame 0 row 0: {c[0, 0].tolist()}")
    print(f"  plain            frame 0 row 0: {ref[0, 0].tolist()}")

# the GPU test's batch-3 case, exactly: golden-derived prompts of 8 rows (left-padded to the longest), the first 3 rows, 20 frames
g = np.load(os.path.join(ROOT, "tests", "golden", "talker_06b_b8.npz"))
lens = [int(x) for x in g["lens"]]
emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(int(g["seed"])), cfg, lens, int(g["n_trail"]), scale=0.05)
out = {}
for flag in ("1", "0"):
    for k in ("QTTS_CP_ATTN_O", "QTTS_CP_FRONT"): _lib.set_option(k, None)
    _lib.set_option("QTTS_CP_ATTN_O", flag)
    eng = TalkerEngine(cfg, wn, weight_dtype=torch.bfloat16, device=dev, max_batch=3, max_seq=256, use_graph=True)
    kw = dict(max_new_tokens=21, min_new_tokens=21, do_sample=False, subtalker_dosample=False, suppress_tokens=sup)
    out[flag] = [eng.generate(emb[:3], mask[:3], tr[:3], pad, **kw).codes.cpu().numpy() for _ in range(2)]
    del eng; torch.cuda.empty_cache()
a, b = out["1"][0], out["0"][0]
print("test-like: shapes", a.shape, b.shape, "fused run-to-run", np.array_equal(out["1"][0], out["1"][1]), "plain run-to-run", np.array_equal(out["0"][0], out["0"][1]))
print("  per frame:", " ".join(f"{(a[:, f] == b[:, f]).mean():.2f}" for f in range(min(a.shape[1], b.shape[1], 8))))
print("  fused frame 0:", a[:, 0].tolist()); print("  plain frame 0:", b[:, 0].tolist())
