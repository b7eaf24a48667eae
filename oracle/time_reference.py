"""TEST / BENCH INFRASTRUCTURE (build container only): time the REFERENCE's own modules on this host's cores.

bench.py's `cpu_baseline` leg runs on the GPU box, where /root/reference does not exist, so it times the oracle port
(kind "port").  This script times the reference's own `nn.Module` code (through oracle/ref_shims.py, seeded random weights at
the metric dims) in the build container and writes profiles/reference_cpu_timing.json, which bench.py attaches to its JSON
line as `cpu_baseline_reference` -- a second, clearly labelled number.

    python oracle/time_reference.py [--frames 24]
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--model", default="1.7b")
    args = ap.parse_args()
    import numpy as np
    import torch
    import gen_golden as gg
    import synth
    nthreads = os.cpu_count()
    torch.set_num_threads(nthreads)
    t = {"1.7b": synth.talker_17b, "0.6b": synth.talker_06b}[args.model]()
    w = synth.talker_weights(t, with_text=False)
    wz = dict(w)
    t_small = synth.TalkerCfg(**{**synth.cfg_dict(t), "text_vocab_size": 8})
    for k, shp in synth.talker_param_shapes(t_small, with_text=True).items():
        if k not in wz:
            wz[k] = np.zeros(shp, np.float32)
    talker = gg.ref_talker(t_small, wz)
    B, F = 8, args.frames
    lens = gg.BENCH_LENS
    emb, mask, trailing, pad = synth.rand_prompt(np.random.default_rng(100), t, lens, 1, scale=0.05)
    model = "n/a"
    try:
        with open("/proc/cpuinfo") as f:
            model = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        pass
    res = {"where": "build container (no GPU)", "cpu_model": model, "cores": nthreads, "threads": nthreads, "kind": "reference",
           "what": "the reference's own modules (qwen_tts/core/models/modeling_qwen3_tts.py talker + nested code_predictor.generate; "
                   "qwen_tts/core/tokenizer_12hz decoder) through oracle/ref_shims.py, greedy HF-4.57.3 loop restated around "
                   "talker.forward, seeded random weights", "runs": []}
    for dt in (torch.float32, torch.bfloat16):
        m = talker if dt == torch.float32 else talker.to(torch.bfloat16)
        if dt == torch.bfloat16:
            for mod in m.modules():
                if hasattr(mod, "rope_init_fn") and hasattr(mod, "inv_freq"):
                    inv, _ = mod.rope_init_fn(mod.config, "cpu")
                    mod.inv_freq = inv
                    mod.original_inv_freq = inv
        e, tr, pd = emb.to(dt), trailing.to(dt), pad.to(dt)
        with torch.no_grad():
            t0 = time.perf_counter()
            gg.restated_sample_loop(m, t, e, mask, tr, pd, max_new_tokens=2, min_new_tokens=2)
            t_pre = time.perf_counter() - t0
            t0 = time.perf_counter()
            codes, _, _ = gg.restated_sample_loop(m, t, e, mask, tr, pd, max_new_tokens=F + 1, min_new_tokens=F + 1)
            t_all = time.perf_counter() - t0
        per_frame = (t_all - t_pre) / (F - 1)
        res["runs"].append({"dtype": str(dt).replace("torch.", ""), "batch": B, "frames": int(codes.shape[1]),
                            "prefill_plus_1_frame_s": round(t_pre, 3), "ms_per_frame": round(1e3 * per_frame, 1),
                            "speech_tokens_per_s": round(B * t.num_code_groups / per_frame, 1),
                            "rtf_x": round(B * 0.08 / per_frame, 3)})
        print(res["runs"][-1], flush=True)
    del talker
    c = synth.codec_real()
    dec = gg.ref_codec_decoder(c, synth.codec_weights(c))
    ac = torch.from_numpy(np.random.default_rng(2).integers(0, c.codebook_size, (B, 125, c.num_quantizers)))
    with torch.no_grad():
        gg.ref_model_decode(dec, c, ac[:1])
        t0 = time.perf_counter()
        gg.ref_model_decode(dec, c, ac)
        t_c = time.perf_counter() - t0
    res["codec_decode_8x10s_fp32_s"] = round(t_c, 2)
    ar = res["runs"][0]
    step_s = ar["prefill_plus_1_frame_s"] + 124 * ar["ms_per_frame"] * 1e-3 + t_c
    res["value"] = round(8 * 125 * 16 / step_s, 1)
    res["unit"] = "speech-tokens/s"
    res["sample"] = (f"metric config (1.7B dims, batch 8, 125 frames + codec decode) extrapolated from prefill + {F} frames fp32 "
                     f"+ the full codec decode: {step_s:.0f} s per step")
    out = os.path.join(ROOT, "profiles", "reference_cpu_timing.json")
    try:
        with open(out) as f:
            allres = json.load(f)
    except Exception:
        allres = {}
    allres[args.model] = res
    with open(out, "w") as f:
        json.dump(allres, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
