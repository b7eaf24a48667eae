#!/usr/bin/env python3
"""In-process, interleaved A/B of engine-level knobs (process-to-process variance on the pool's boxes is ~5 %, larger than the
effects being measured): for each repetition, for each configuration: set the switches through the C ABI (qtts_set_option), build a
1.7B bf16 talker engine (engine-level switches are copied at engine construction), time `--frames` frame steps (min of 3), destroy it.

    python tools/ab_inproc.py --frames 40 --reps 3
"""
import argparse, gc, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import synth
from qwen3_tts_amd import _lib
from qwen3_tts_amd.talker import TalkerEngine

CONFIGS = {
    "default": {},
    "cp_layer_off": {"QTTS_CP_LAYER": "0"},      # round 6: the layer as its two fused launches (round 5's frame step)
    "layer_gu_entry": {"QTTS_CP_LAYER_GU_WHEN": "0"}, "layer_gu_wo": {"QTTS_CP_LAYER_GU_WHEN": "1"},        # the gate|up block's LDS-DMA at entry / behind the o-projection operator's requests (default: behind the attention stage)
    "layer_hid0": {"QTTS_CP_LAYER_HID_MODE": "0"}, "layer_hid2": {"QTTS_CP_LAYER_HID_MODE": "2"},           # hidden rows: every wave polls its whole quarter / sentinels + a read the L2 may serve (default 1: sentinels + one sc1 read)
    "layer_pace2": {"QTTS_CP_LAYER_GU_PACE": "2"}, "layer_pace4": {"QTTS_CP_LAYER_GU_PACE": "4"}, "layer_pace8": {"QTTS_CP_LAYER_GU_PACE": "8"},       # x 64 clocks between the gate|up block's DMA requests (default 0: back to back)
    "layer_h8": {"QTTS_CP_LAYER_PAUSE_H": "8"}, "layer_h12": {"QTTS_CP_LAYER_PAUSE_H": "12"}, "layer_h20": {"QTTS_CP_LAYER_PAUSE_H": "20"},
    "layer_h24": {"QTTS_CP_LAYER_PAUSE_H": "24"}, "layer_h32": {"QTTS_CP_LAYER_PAUSE_H": "32"}, "layer_h4": {"QTTS_CP_LAYER_PAUSE_H": "4"},
    "cp_mlp_off": {"QTTS_CP_MLP": "0"},          # round 5: the code predictor's MLP as two decode GEMMs (round 4's frame step)
    "f32_fused_mlp": {"QTTS_CP_MLP_F32": "1"},
    "f32_fused_both": {"QTTS_CP_MLP_F32": "1", "QTTS_CP_ATTN_O_F32": "1"},     # ... and cp_attn_o_kernel<.., .., true>: every launch of passes >= 1 fused       # --dtype f32: cp_mlp_kernel<true, ...> against the fp32 split-K plan (the default there)
    "cp_fused_off": {"QTTS_CP_MLP": "0", "QTTS_CP_ATTN_O": "0"},   # ... and q|k|v / attention / o-projection as separate launches (round 3's)
    "mlp_b16_c24": {"QTTS_CP_MLP_PAUSE_B": "16"}, "mlp_b20_c24": {"QTTS_CP_MLP_PAUSE_B": "20"}, "mlp_b28_c24": {"QTTS_CP_MLP_PAUSE_B": "28"},
    "mlp_b24_c16": {"QTTS_CP_MLP_PAUSE_C": "16"}, "mlp_b24_c20": {"QTTS_CP_MLP_PAUSE_C": "20"}, "mlp_b24_c28": {"QTTS_CP_MLP_PAUSE_C": "28"},
    "mlp_b24_c32": {"QTTS_CP_MLP_PAUSE_C": "32"}, "mlp_b28_c28": {"QTTS_CP_MLP_PAUSE_B": "28", "QTTS_CP_MLP_PAUSE_C": "28"},
    "mlp_step2": {"QTTS_CP_MLP_STEP": "2"}, "mlp_step8": {"QTTS_CP_MLP_STEP": "8"},
    "ao_pause20": {"QTTS_CP_ATTN_O_PAUSE": "20"}, "ao_pause24": {"QTTS_CP_ATTN_O_PAUSE": "24"}, "ao_pause12": {"QTTS_CP_ATTN_O_PAUSE": "12"},
    "ao_step2": {"QTTS_CP_ATTN_O_STEP": "2"}, "ao_step6": {"QTTS_CP_ATTN_O_STEP": "6"}, "ao_step8": {"QTTS_CP_ATTN_O_STEP": "8"},
    "skinny8_nw8": {"QTTS_SKINNY8_NW": "8"},
    "skinny8_nw4": {"QTTS_SKINNY8_NW": "4"},
    # round 6, --batch 32: the o- / down-projections at batch 17..32 with K split over workgroups (skinny2_ks_kernel)
    "mlp32_off": {"QTTS_CP_MLP32": "0"}, "mlp32_off_ks_off": {"QTTS_CP_MLP32": "0", "QTTS_SKINNY_KS": "0"},       # round 6, --batch 32: the code predictor's MLP as one launch at batch 9..32
    "mlp32_b16": {"QTTS_CP_MLP_PAUSE_B": "16"}, "mlp32_b32": {"QTTS_CP_MLP_PAUSE_B": "32"}, "mlp32_b40": {"QTTS_CP_MLP_PAUSE_B": "40"}, "mlp32_c32": {"QTTS_CP_MLP_PAUSE_C": "32"}, "mlp32_c40": {"QTTS_CP_MLP_PAUSE_C": "40"},
    "ks_off": {"QTTS_SKINNY_KS": "0"}, "ks_mink2048": {"QTTS_SKINNY_KS_MINK": "2048"}, "ks_mink6144": {"QTTS_SKINNY_KS_MINK": "6144"},
    "ks_pause0": {"QTTS_SKINNY_KS_PAUSE": "0"}, "ks_pause16": {"QTTS_SKINNY_KS_PAUSE": "16"}, "ks_pause24": {"QTTS_SKINNY_KS_PAUSE": "24"},
    "ks_pause32": {"QTTS_SKINNY_KS_PAUSE": "32"}, "ks_pause48": {"QTTS_SKINNY_KS_PAUSE": "48"}, "ks_pause64": {"QTTS_SKINNY_KS_PAUSE": "64"},
    "ks_mink6144_p32": {"QTTS_SKINNY_KS_MINK": "6144", "QTTS_SKINNY_KS_PAUSE": "32"}, "ks_mink6144_p48": {"QTTS_SKINNY_KS_MINK": "6144", "QTTS_SKINNY_KS_PAUSE": "48"},
}
KEYS = sorted({k for c in CONFIGS.values() for k in c})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40); ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--batch", type=int, default=8, help="rows of the frame step (32: BASELINE configs 4 / 5)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"], help="f32: the exact parity mode (fp32 operators; cp_mlp_kernel<true, ...> against the split-K plan)")
    a = ap.parse_args()
    t = synth.talker_17b()
    base = np.random.default_rng(0).standard_normal(1 << 20, dtype=np.float32)
    w = {}
    for k, shp in synth.talker_param_shapes(t, with_text=False).items():
        v = np.resize(base, int(np.prod(shp))).reshape(shp) * np.float32(0.08 if "head" in k else 0.02)
        if "norm" in k and k.endswith("weight"): v = v * 0 + 1
        if "head" in k: v = np.random.default_rng(__import__("zlib").crc32(k.encode())).standard_normal(shp, dtype=np.float32) * np.float32(0.08)   # no tied logits
        w[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    B, F = a.batch, a.frames
    lens = [24 + 4 * (i % 8) + 12 for i in range(B)]
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(1), t, lens, 1)
    sup = [i for i in range(t.vocab_size - 1024, t.vocab_size) if i != t.codec_eos_token_id]
    kw = dict(max_new_tokens=F + 1, min_new_tokens=F + 1, suppress_tokens=sup, output_hidden_states=False)
    names = [n for n in CONFIGS if not a.only or n in a.only]
    res = {n: [] for n in names}
    for rep in range(a.reps):
        for n in names:
            for k in KEYS:
                _lib.set_option(k, CONFIGS[n].get(k))
            eng = TalkerEngine(t, w, weight_dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32, max_batch=B, max_seq=64 + F + 8, use_graph=True)
            eng.generate(emb, mask, tr, pad, seed=0, **kw); torch.cuda.synchronize()
            ts = []
            for r in range(3):
                t1 = time.perf_counter(); eng.generate(emb, mask, tr, pad, seed=r, **kw); torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            t1 = time.perf_counter(); eng.generate(emb, mask, tr, pad, seed=0, **dict(kw, max_new_tokens=1, min_new_tokens=1)); torch.cuda.synchronize()
            tp = time.perf_counter() - t1
            ms = 1000 * (min(ts) - tp) / F
            res[n].append(round(ms, 4))
            st = eng.stats()
            print(f"[ab_inproc] rep {rep} {n:28s} {ms:.3f} ms/frame   (graph nodes {st['graph_nodes']}, fused launches per step: attention {st['cp_fused_per_step']}, "
                  f"mlp {st['cp_mlp_per_step']}, whole layer {st.get('cp_layer_per_step', 0)}; split-K GEMMs {st.get('ks_split_per_step', 0)})", flush=True)
            del eng; gc.collect(); torch.cuda.empty_cache()
    out = {n: {"ms_per_frame": v, "min": min(v), "median": float(np.median(v))} for n, v in res.items()}
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", ("ab_inproc.json" if a.batch == 8 else f"ab_inproc_b{a.batch}.json") if a.dtype == "bf16" else "ab_inproc_f32.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
