"""Diagnostic (not a test): run the HIP path against the oracle/goldens on tiny configs and print diffs.
    gpurun -- python tools/diag_parity.py [codec] [talker] [real]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import synth, codec_ref, talker_ref
from qwen3_tts_amd.codec import CodecDecoderEngine
from qwen3_tts_amd.talker import TalkerEngine

G = os.path.join(ROOT, "tests", "golden")
what = set(sys.argv[1:]) or {"codec", "talker"}

def td(w): return {k: torch.from_numpy(v) for k, v in w.items()}

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max()), float(np.sqrt(((a - b) ** 2).mean())), float(np.sqrt((b ** 2).mean()))

if "codec" in what:
    c = synth.codec_tiny(); w = synth.codec_weights(c); g = np.load(os.path.join(G, "codec_tiny.npz"))
    for dt in (torch.float32, torch.bfloat16):
        eng = CodecDecoderEngine(c, td(w), compute_dtype=dt, max_batch=3, max_frames=64)
        codes = torch.from_numpy(g["fwd_codes"]).cuda()
        for st, key in [("rvq", "fwd_rvq"), ("pre_conv", "fwd_pre_conv"), ("pre_transformer", "fwd_pre_transformer_btc"),
                        ("upsample0", "fwd_upsample0"), ("upsample1", "fwd_upsample1"), ("decoder0", "fwd_decoder0"),
                        ("block1", "fwd_block1"), ("block2", "fwd_block2"), ("block3", "fwd_block3"), ("block4", "fwd_block4")]:
            y = eng.forward_stage(codes, st).cpu().numpy()
            ref = g[key]
            if key != "fwd_pre_transformer_btc": ref = ref.transpose(0, 2, 1)
            print(f"[codec {dt}] {st:16s} shape {y.shape} max|d| %.3e rms(d) %.3e rms(ref) %.3e" % rel(y, ref))
        wav, pre = eng.forward(codes, return_pre_clamp=True)
        print(f"[codec {dt}] wav              max|d| %.3e rms(d) %.3e rms(ref) %.3e" % rel(wav.cpu().numpy(), g["fwd_wav"]))
        print(f"[codec {dt}] pre_clamp        max|d| %.3e rms(d) %.3e rms(ref) %.3e" % rel(pre.cpu().numpy(), g["fwd_pre_clamp"]))
        cw = eng.chunked_decode(torch.from_numpy(g["chunk_codes"]).cuda(), 16, 5)
        print(f"[codec {dt}] chunk(16,5)      max|d| %.3e rms(d) %.3e rms(ref) %.3e" % rel(cw.cpu().numpy(), g["chunk_wav_16_5"]))
        cw = eng.chunked_decode(torch.from_numpy(g["chunk_codes"]).cuda())
        print(f"[codec {dt}] chunk(default)   max|d| %.3e rms(d) %.3e rms(ref) %.3e" % rel(cw.cpu().numpy(), g["chunk_wav_default"]))
        wv, lens = eng.decode_padded(torch.from_numpy(g["ragged_codes"]).cuda())
        for i in range(3):
            print(f"[codec {dt}] ragged{i} len {lens[i]} max|d| %.3e rms(d) %.3e rms(ref) %.3e" % rel(wv[i, :lens[i]].cpu().numpy(), g[f"ragged_wav{i}"]))
        del eng

if "talker" in what:
    t = synth.talker_tiny(); w = synth.talker_weights(t); g = np.load(os.path.join(G, "talker_tiny.npz"))
    sup = [i for i in range(t.vocab_size - 1024, t.vocab_size) if i != t.codec_eos_token_id]
    for dt, graph in ((torch.bfloat16, False), (torch.bfloat16, True), (torch.float32, True)):
        eng = TalkerEngine(t, td(w), weight_dtype=dt, max_batch=4, max_seq=256, use_graph=graph)
        out = eng.generate(torch.from_numpy(g["embeds"]), torch.from_numpy(g["mask"]), torch.from_numpy(g["trailing"]),
                           torch.from_numpy(g["tts_pad"]), max_new_tokens=14, min_new_tokens=2, do_sample=False,
                           subtalker_dosample=False, repetition_penalty=1.05, suppress_tokens=sup)
        codes = out.codes.cpu().numpy(); toks = out.tokens.cpu().numpy()
        print(f"[talker {dt} graph={graph}] n_frames {out.n_frames} (golden {g['codes'].shape[1]}) stats {eng.stats()}")
        n = min(codes.shape[1], g["codes"].shape[1])
        print("   tokens match:", (toks[:, :n + 1] == g["tokens"][:, :n + 1]).mean(), " codes match:", (codes[:, :n] == g["codes"][:, :n]).mean())
        print("   tokens:", toks.tolist())
        print("   golden:", g["tokens"].tolist())
        if out.hidden is not None and n > 0:
            print("   hidden max|d| %.3e rms(d) %.3e rms(ref) %.3e" % rel(out.hidden[:, :n].cpu().numpy(), g["hidden"][:, :n]))
        out2 = eng.generate(torch.from_numpy(g["embeds"]), torch.from_numpy(g["mask"]), torch.from_numpy(g["trailing"]),
                            torch.from_numpy(g["tts_pad"]), max_new_tokens=14, min_new_tokens=2, do_sample=False,
                            subtalker_dosample=False, repetition_penalty=1.05, suppress_tokens=sup, eos_token_id=int(g["eos2"]))
        print("   eos2 run: n_frames", out2.n_frames, "golden", g["codes_eos2"].shape[1], "tokens", out2.tokens.cpu().numpy().tolist())
        if out2.n_frames == g["codes_eos2"].shape[1]:
            print("   eos2 codes match:", (out2.codes.cpu().numpy() == g["codes_eos2"]).mean())
        # sampling smoke
        out3 = eng.generate(torch.from_numpy(g["embeds"]), torch.from_numpy(g["mask"]), torch.from_numpy(g["trailing"]),
                            torch.from_numpy(g["tts_pad"]), max_new_tokens=10, suppress_tokens=sup, seed=5)
        print("   sampled tokens:", out3.tokens.cpu().numpy().tolist())
        del eng
print("diag done")
