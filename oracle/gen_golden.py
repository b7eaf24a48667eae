"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Generates tests/golden/*.npz by running the UNMODIFIED reference modules
(/root/reference/qwen_tts, via oracle/ref_shims.py) on seeded synthetic weights
(synth.py at the repo root).  Runs only in the build container (the reference tree is absent
on the GPU box); the fixtures it writes are committed.

    python oracle/gen_golden.py [--only codec_tiny,codec_real,talker_tiny,talker_06b,talker_17b,prompt_tiny]

What drives what:
  * codec:  Qwen3TTSTokenizerV2Decoder.forward / .chunked_decode (V2:869-896) and the body of
            Qwen3TTSTokenizerV2Model.decode (V2:993-1024) called unbound on a stand-in `self`
            (so the Mimi encoder never has to be built).
  * talker: Qwen3TTSTalkerForConditionalGeneration.forward (M:1636-1744) incl. its nested
            `code_predictor.generate` (runs unmodified), driven by a hand-written HF-4.57.3
            `_sample` loop (the installed transformers 5.x outer loop no longer passes
            `cache_position`) -- `restated_sample_loop` below.
  * prompt: Qwen3TTSForConditionalGeneration.generate (M:2022-2292) with `talker.generate`
            replaced by a recorder, so the embeddings / mask / trailing text it assembles are pinned.
"""
import argparse
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))      # synth.py (synthetic weights) lives at the repo root
import ref_shims  # noqa: E402
import synth  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _load(module, w, prefix=""):
    import torch
    sd = module.state_dict()
    new = {}
    for k in sd:
        kk = k[len(prefix):] if prefix and k.startswith(prefix) else k
        if kk in w:
            new[k] = torch.from_numpy(w[kk])
    missing = [k for k in sd if k not in new]
    assert not missing, f"synthetic weights do not cover: {missing[:8]}"
    module.load_state_dict(new, strict=True)


# ----------------------------------------------------------------------------- codec
def ref_codec_decoder(c: synth.CodecCfg, w):
    ref_shims.install()
    from qwen_tts.core.tokenizer_12hz.configuration_qwen3_tts_tokenizer_v2 import Qwen3TTSTokenizerV2DecoderConfig
    from qwen_tts.core.tokenizer_12hz.modeling_qwen3_tts_tokenizer_v2 import Qwen3TTSTokenizerV2Decoder
    cfg = Qwen3TTSTokenizerV2DecoderConfig(
        codebook_size=c.codebook_size, codebook_dim=c.codebook_dim, hidden_size=c.hidden_size,
        latent_dim=c.latent_dim, num_attention_heads=c.num_attention_heads,
        num_key_value_heads=c.num_key_value_heads, head_dim=c.head_dim, sliding_window=c.sliding_window,
        intermediate_size=c.intermediate_size, num_hidden_layers=c.num_hidden_layers,
        num_quantizers=c.num_quantizers, upsample_rates=c.upsample_rates,
        upsampling_ratios=c.upsampling_ratios, decoder_dim=c.decoder_dim, rms_norm_eps=c.rms_norm_eps,
        rope_theta=c.rope_theta, max_position_embeddings=c.max_position_embeddings)
    cfg._attn_implementation = "eager"
    m = Qwen3TTSTokenizerV2Decoder(cfg).eval()
    _load(m, w)
    return m


def ref_model_decode(dec, c, audio_codes):
    """Run the reference's Qwen3TTSTokenizerV2Model.decode body (V2:993-1024) on a stand-in self."""
    from qwen_tts.core.tokenizer_12hz.modeling_qwen3_tts_tokenizer_v2 import Qwen3TTSTokenizerV2Model
    fake = types.SimpleNamespace(config=types.SimpleNamespace(return_dict=True),
                                 decode_upsample_rate=c.total_upsample, decoder=dec)
    return Qwen3TTSTokenizerV2Model.decode(fake, audio_codes).audio_values


def gen_codec_tiny():
    import torch
    c = synth.codec_tiny()
    w = synth.codec_weights(c)
    dec = ref_codec_decoder(c, w)
    g = np.random.default_rng(11)
    out = {"weights_checksum": synth.weights_checksum(w)}
    # (a) single forward, all stage outputs (hooks on the reference's own submodules)
    codes = torch.from_numpy(g.integers(0, c.codebook_size, (1, c.num_quantizers, 6)))   # small: every stage is stored
    stages = {}
    hooks = []

    def hook(name):
        return lambda mod, inp, o: stages.__setitem__(name, (o.last_hidden_state if hasattr(o, "last_hidden_state") else o).detach().clone())
    hooks.append(dec.pre_conv.register_forward_hook(hook("pre_conv")))
    hooks.append(dec.pre_transformer.register_forward_hook(hook("pre_transformer_btc")))
    for u in range(len(c.upsampling_ratios)):
        hooks.append(dec.upsample[u][1].register_forward_hook(hook(f"upsample{u}")))
    hooks.append(dec.decoder[0].register_forward_hook(hook("decoder0")))
    for i in range(len(c.upsample_rates)):
        hooks.append(dec.decoder[i + 1].register_forward_hook(hook(f"block{i + 1}")))
    hooks.append(dec.decoder[-1].register_forward_hook(hook("pre_clamp")))
    with torch.no_grad():
        rvq = dec.quantizer.decode(codes)
        wav = dec(codes)
    for h in hooks:
        h.remove()
    out["fwd_codes"] = codes.numpy()
    out["fwd_rvq"] = rvq.numpy()
    out["fwd_wav"] = wav.numpy()
    for k, v in stages.items():
        out["fwd_" + k] = v.numpy()
    # (b) chunked decode with a small chunk so several boundaries are crossed (V2:886-896 takes the sizes as args)
    codes2 = torch.from_numpy(g.integers(0, c.codebook_size, (2, c.num_quantizers, 41)))
    with torch.no_grad():
        out["chunk_codes"] = codes2.numpy()
        out["chunk_wav_16_5"] = dec.chunked_decode(codes2, chunk_size=16, left_context_size=5).numpy()
        out["chunk_wav_default"] = dec.chunked_decode(codes2).numpy()
    # (c) model.decode on a ragged, -1 padded batch (V2:993-1024)
    lens = [13, 4, 19]
    ac = torch.full((3, max(lens), c.num_quantizers), -1, dtype=torch.long)
    for i, l in enumerate(lens):
        ac[i, :l] = torch.from_numpy(g.integers(0, c.codebook_size, (l, c.num_quantizers)))
    with torch.no_grad():
        wavs = ref_model_decode(dec, c, ac)
    out["ragged_codes"] = ac.numpy()
    for i, a in enumerate(wavs):
        out[f"ragged_wav{i}"] = a.numpy()
    np.savez_compressed(os.path.join(GOLDEN, "codec_tiny.npz"), **out)
    print("codec_tiny:", {k: getattr(v, "shape", v) for k, v in out.items()})


def gen_codec_real():
    import torch
    c = synth.codec_real()
    w = synth.codec_weights(c)
    dec = ref_codec_decoder(c, w)
    g = np.random.default_rng(11)
    out = {"weights_checksum": synth.weights_checksum(w)}
    # BASELINE config 2: 10 s = 125 frames of random codes, B=1
    codes = torch.from_numpy(g.integers(0, c.codebook_size, (1, 125, c.num_quantizers)))
    t = time.time()
    with torch.no_grad():
        wav = ref_model_decode(dec, c, codes)[0]
    out["t125_seconds_ref_cpu"] = time.time() - t
    out["t125_codes"] = codes.numpy().astype(np.int16)
    out["t125_wav"] = wav.numpy()
    # two-chunk path (325 frames > 300): keep the region around the chunk seam + a strided sample
    codes2 = torch.from_numpy(g.integers(0, c.codebook_size, (1, 325, c.num_quantizers)))
    with torch.no_grad():
        wav2 = ref_model_decode(dec, c, codes2)[0].numpy()
    seam = 300 * c.total_upsample
    out["t325_codes"] = codes2.numpy().astype(np.int16)
    out["t325_seam_lo"] = seam - 4096
    out["t325_seam"] = wav2[seam - 4096: seam + 8192]
    out["t325_stride"] = 61
    out["t325_strided"] = wav2[::61]
    out["t325_sum"] = float(wav2.astype(np.float64).sum())
    out["t325_len"] = wav2.shape[0]
    np.savez_compressed(os.path.join(GOLDEN, "codec_real.npz"), **out)
    print("codec_real: ref cpu 125 frames %.2fs" % out["t125_seconds_ref_cpu"], wav.shape, wav2.shape)


# ----------------------------------------------------------------------------- talker
def ref_talker_cfgs(t: synth.TalkerCfg):
    ref_shims.install()
    from qwen_tts.core.models.configuration_qwen3_tts import Qwen3TTSTalkerConfig, Qwen3TTSConfig
    hd2 = t.head_dim // 2
    sec = [hd2 - 2 * (hd2 // 3), hd2 // 3, hd2 // 3]
    cp = dict(vocab_size=t.cp_vocab_size, hidden_size=t.cp_hidden_size, intermediate_size=t.cp_intermediate_size,
              num_hidden_layers=t.cp_num_hidden_layers, num_attention_heads=t.cp_num_attention_heads,
              num_key_value_heads=t.cp_num_key_value_heads, head_dim=t.cp_head_dim,
              rms_norm_eps=t.cp_rms_norm_eps, rope_theta=t.cp_rope_theta, num_code_groups=t.num_code_groups,
              pad_token_id=None)
    tk = dict(code_predictor_config=cp, vocab_size=t.vocab_size, hidden_size=t.hidden_size,
              intermediate_size=t.intermediate_size, num_hidden_layers=t.num_hidden_layers,
              num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
              head_dim=t.head_dim, rms_norm_eps=t.rms_norm_eps, rope_theta=t.rope_theta,
              rope_scaling={"rope_type": "default", "mrope_section": sec, "interleaved": True},
              num_code_groups=t.num_code_groups, text_hidden_size=t.text_hidden_size,
              text_vocab_size=t.text_vocab_size, codec_eos_token_id=t.codec_eos_token_id,
              codec_think_id=t.codec_think_id, codec_nothink_id=t.codec_nothink_id,
              codec_think_bos_id=t.codec_think_bos_id, codec_think_eos_id=t.codec_think_eos_id,
              codec_pad_id=t.codec_pad_id, codec_bos_id=t.codec_bos_id, spk_id=t.spk_id,
              spk_is_dialect=t.spk_is_dialect, codec_language_id=t.codec_language_id, pad_token_id=None)
    return Qwen3TTSTalkerConfig, Qwen3TTSConfig, tk


def ref_talker(t: synth.TalkerCfg, w):
    ref_shims.install()
    from qwen_tts.core.models.modeling_qwen3_tts import Qwen3TTSTalkerForConditionalGeneration  # noqa
    TalkerConfig, _, tk = ref_talker_cfgs(t)
    from qwen_tts.core.models.modeling_qwen3_tts import Qwen3TTSTalkerForConditionalGeneration
    cfg = TalkerConfig(**tk)
    cfg._attn_implementation = "eager"
    cfg.code_predictor_config._attn_implementation = "eager"
    import torch
    with torch.device("meta"):
        m = Qwen3TTSTalkerForConditionalGeneration(cfg)
    m = m.to_empty(device="cpu").eval()
    # non-persistent rotary buffers are lost by to_empty: rebuild them
    for mod in m.modules():
        if hasattr(mod, "rope_init_fn") and hasattr(mod, "inv_freq"):
            inv, _ = mod.rope_init_fn(mod.config, "cpu")
            mod.inv_freq = inv
            mod.original_inv_freq = inv
    _load(m, w)
    m.config._attn_implementation = "eager"
    return m


def restated_sample_loop(talker, t: synth.TalkerCfg, embeds, mask, trailing, tts_pad, max_new_tokens,
                         min_new_tokens=2, eos_token_id=None, repetition_penalty=1.05, trace=None):
    """Greedy HF-4.57.3 `_sample` around the REFERENCE talker.forward.
    Processors restated from transformers generation/logits_process.py in `_get_logits_processor`
    order: RepetitionPenalty -> MinNewTokensLength -> SuppressTokens; then argmax."""
    import torch
    eos = t.codec_eos_token_id if eos_token_id is None else eos_token_id
    suppress = [i for i in range(t.vocab_size - 1024, t.vocab_size) if i != t.codec_eos_token_id]
    B, T, _ = embeds.shape
    talker.rope_deltas = None
    o = talker(inputs_embeds=embeds, attention_mask=mask, use_cache=True, output_hidden_states=True,
               trailing_text_hidden=trailing, tts_pad_embed=tts_pad)
    generated = torch.zeros(B, 0, dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)
    frames, hiddens = [], []
    step = 0
    while True:
        s = o.logits[:, -1].float().clone()
        if trace is not None:
            trace.setdefault("logits", []).append(s.clone())
        if generated.shape[1] > 0:
            sc = torch.gather(s, 1, generated)
            sc = torch.where(sc < 0, sc * repetition_penalty, sc / repetition_penalty)
            s = s.scatter(1, generated, sc)
        if generated.shape[1] < min_new_tokens:
            s[:, eos] = float("-inf")
        s[:, suppress] = float("-inf")
        if trace is not None:
            top2 = torch.topk(s, 2, dim=-1)[0]
            trace.setdefault("margin", []).append((top2[:, 0] - top2[:, 1]).clone())
        tok = torch.argmax(s, dim=-1)
        tok = tok * unfinished + eos * (1 - unfinished)
        generated = torch.cat((generated, tok[:, None]), dim=1)
        unfinished = unfinished & (tok != eos).long()
        if generated.shape[1] >= max_new_tokens or unfinished.max() == 0:
            break
        mask = torch.cat([mask, mask.new_ones(B, 1)], 1)
        hiddens.append(o.past_hidden[:, 0])
        o = talker(input_ids=tok[:, None], attention_mask=mask, past_key_values=o.past_key_values, use_cache=True,
                   cache_position=torch.tensor([T + step]), past_hidden=o.past_hidden,
                   generation_step=o.generation_step, trailing_text_hidden=trailing, tts_pad_embed=tts_pad,
                   output_hidden_states=True, subtalker_dosample=False, subtalker_top_k=None,
                   subtalker_top_p=None, subtalker_temperature=None)
        frames.append(o.hidden_states[1])
        step += 1
    codes = torch.stack(frames, dim=1) if frames else torch.zeros(B, 0, t.num_code_groups, dtype=torch.long)
    hidden = torch.stack(hiddens, dim=1) if hiddens else torch.zeros(B, 0, embeds.shape[-1])
    return codes, generated, hidden


_rand_prompt = synth.rand_prompt


def gen_talker_tiny():
    import torch
    t = synth.talker_tiny()
    w = synth.talker_weights(t)
    talker = ref_talker(t, w)
    g = np.random.default_rng(21)
    out = {"weights_checksum": synth.weights_checksum(w)}
    emb, mask, trailing, pad = _rand_prompt(g, t, [9, 14, 5], 4, scale=0.5)
    with torch.no_grad():
        tr = {}
        codes, toks, hidden = restated_sample_loop(talker, t, emb, mask, trailing, pad, max_new_tokens=14, trace=tr)
        # EOS handling: re-run with eos := row 0's 6th token so that row finishes early while others go on
        eos2 = int(toks[0, 5])
        codes2, toks2, hidden2 = restated_sample_loop(talker, t, emb, mask, trailing, pad, max_new_tokens=14,
                                                      eos_token_id=eos2)
    out.update(embeds=emb.numpy(), mask=mask.numpy(), trailing=trailing.numpy(), tts_pad=pad.numpy(),
               codes=codes.numpy(), tokens=toks.numpy(), hidden=hidden.numpy(),
               logits=torch.stack(tr["logits"], 1).numpy(), margin=torch.stack(tr["margin"], 1).numpy(),
               eos2=eos2, codes_eos2=codes2.numpy(), tokens_eos2=toks2.numpy())
    np.savez_compressed(os.path.join(GOLDEN, "talker_tiny.npz"), **out)
    print("talker_tiny: codes", codes.shape, "eos2", eos2, "tokens_eos2", toks2.tolist())


def _gen_talker_real(name, t, lens, n_trail, max_new, seed, min_new=2, logit_steps=None):
    import torch
    w = synth.talker_weights(t, with_text=False)
    t0 = time.time()
    # the reference module still owns text_embedding / text_projection parameters: fill the unused ones with zeros
    shapes = synth.talker_param_shapes(t, with_text=True)
    wz = dict(w)
    # keep RAM bounded: a 1-row stand-in cannot be loaded strictly, so build the module with a small text vocab
    t_small = synth.TalkerCfg(**{**synth.cfg_dict(t), "text_vocab_size": 8})
    for k, shp in synth.talker_param_shapes(t_small, with_text=True).items():
        if k not in wz:
            wz[k] = np.zeros(shp, np.float32)
    talker = ref_talker(t_small, wz)
    g = np.random.default_rng(seed)
    emb, mask, trailing, pad = _rand_prompt(g, t, lens, n_trail, scale=0.05)
    tr = {}
    t1 = time.time()
    with torch.no_grad():
        codes, toks, hidden = restated_sample_loop(talker, t, emb, mask, trailing, pad, max_new_tokens=max_new,
                                                   min_new_tokens=min_new, trace=tr)
    dt = time.time() - t1
    out = {"weights_checksum": synth.weights_checksum(w), "lens": np.array(lens), "n_trail": n_trail,
           "seed": seed, "max_new": max_new, "min_new": min_new, "codes": codes.numpy(), "tokens": toks.numpy(),
           "margin": torch.stack(tr["margin"], 1).numpy(), "logits0": tr["logits"][0].numpy(),
           "hidden_last": hidden[:, -1].numpy(), "ref_cpu_seconds": dt,
           "ref_cpu_threads": torch.get_num_threads()}
    hs = [f for f in (0, 3, 7, 15, 23, 31, 39, 63, 95) if f < hidden.shape[1]]      # hidden states of selected frames: a run that a low-margin
    out["hidden_steps"] = np.array(hs)                                              # flip shortens is still compared up to where it agrees
    out["hidden_sel"] = torch.stack([hidden[:, f] for f in hs], 0).numpy()
    if logit_steps is not None:      # raw cb-0 logits (before the HF processors) of selected token steps, for teacher-forced comparisons
        out["logit_steps"] = np.array(logit_steps)
        out["logits_sel"] = torch.stack([tr["logits"][i] for i in logit_steps], 0).numpy()
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **out)
    print(f"{name}: codes {tuple(codes.shape)} ref cpu loop {dt:.1f}s (build {t1 - t0:.1f}s) min margin {out['margin'].min():.5f}")


def gen_talker_06b():
    # BASELINE config 1: 0.6B dims, 1 utterance, greedy, 64 new tokens (-> 63 frames)
    _gen_talker_real("talker_06b", synth.talker_06b(), [61], 1, 64, 7)


def gen_talker_17b():
    # bench dims (1.7B), ragged batch of 3, short
    _gen_talker_real("talker_17b", synth.talker_17b(), [40, 52, 33], 6, 20, 8)


def gen_talker_06b_b8():
    # BASELINE config 3: 0.6B dims, batch 8, ragged prompts (text 24..52 ids + 12 role/codec-prefix rows), length forced
    # to 125 frames (10 s) with min_new_tokens = max_new_tokens = 126, greedy
    _gen_talker_real("talker_06b_b8", synth.talker_06b(), [36, 40, 44, 48, 52, 56, 60, 64], 1, 126, 9, min_new=126)


def gen_talker_17b_b32():
    # BASELINE config 4 shape: 1.7B dims, batch 32, streaming text input (24 trailing text rows fed one per frame,
    # M:2229-2232), ragged prompts, greedy, 43 frames: the trailing text runs out at frame 24 and the remaining 19 frames take
    # the tts_pad branch (M:1689-1692) at batch 32 / real dims (round 2's fixture stopped at 12 frames)
    _gen_talker_real("talker_17b_b32", synth.talker_17b(), [40 + (7 * i) % 32 for i in range(32)], 24, 44, 10)


BENCH_LENS = [36, 40, 44, 48, 52, 56, 60, 64]          # bench.py: 24 + 4 * (i % 8) + 12
LOGIT_STEPS = [0, 1, 2, 3, 4, 8, 16, 32, 48, 64, 96, 125]


def gen_talker_17b_b8():
    # THE METRIC CONFIG (BASELINE.json metric): 1.7B dims, batch 8, bench.py's ragged prompts, 125 frames (10 s) per utterance
    # forced with min_new_tokens = max_new_tokens = 126, greedy, fp32 reference CPU path
    _gen_talker_real("talker_17b_b8", synth.talker_17b(), BENCH_LENS, 1, 126, 100, min_new=126, logit_steps=LOGIT_STEPS)


def gen_talker_06b_long():
    # a LONG utterance (reference default max_new_tokens = 2048, IM:329): 0.6B dims, ragged batch of 2, 820 frames (65.6 s) forced,
    # greedy fp32 -- the KV cache grows to ~865 keys, far past the decode attention's 256-key register window
    _gen_talker_real("talker_06b_long", synth.talker_06b(), [30, 45], 40, 821, 11, min_new=821)


def gen_talker_17b_b8_bf16():
    """The reference's own modules in **bfloat16** (the dtype of its examples, examples/test_model_12hz_*.py), TEACHER-FORCED
    frame by frame with the fp32 golden above: every frame's talker input is the fp32 golden's 16 codes (the nested
    code_predictor.generate runs free inside the frame -- its own 15 codes are recorded -- and its result is replaced by the
    golden's before the embedding sum), the fed cb-0 token is the golden's.  Recorded: the bf16 reference's own greedy choice
    for all 16 codebooks of every frame and its raw cb-0 logits at LOGIT_STEPS.  This is the yardstick for the MI355X bf16
    engine: how far does bf16 move the reference itself, and is the engine inside that distance?"""
    import torch
    t = synth.talker_17b()
    g = np.load(os.path.join(GOLDEN, "talker_17b_b8.npz"))
    w = synth.talker_weights(t, with_text=False)
    assert abs(synth.weights_checksum(w) - float(g["weights_checksum"])) < 1e-3 * max(1.0, abs(float(g["weights_checksum"])))
    wz = dict(w)
    t_small = synth.TalkerCfg(**{**synth.cfg_dict(t), "text_vocab_size": 8})
    for k, shp in synth.talker_param_shapes(t_small, with_text=True).items():
        if k not in wz:
            wz[k] = np.zeros(shp, np.float32)
    talker = ref_talker(t_small, wz).to(torch.bfloat16)
    for mod in talker.modules():          # rotary inv_freq stays fp32 in the reference (non-persistent buffer, computed in fp32)
        if hasattr(mod, "rope_init_fn") and hasattr(mod, "inv_freq"):
            inv, _ = mod.rope_init_fn(mod.config, "cpu")
            mod.inv_freq = inv
            mod.original_inv_freq = inv
    lens = [int(x) for x in g["lens"]]
    rng = np.random.default_rng(int(g["seed"]))
    emb, mask, trailing, pad = _rand_prompt(rng, t, lens, int(g["n_trail"]), scale=0.05)
    emb, trailing, pad = emb.to(torch.bfloat16), trailing.to(torch.bfloat16), pad.to(torch.bfloat16)
    gold_codes = torch.from_numpy(g["codes"])          # (B, F, 16)
    gold_tokens = torch.from_numpy(g["tokens"])        # (B, F + 1)
    B, F, G = gold_codes.shape
    own_sub, frame_no = [], [0]
    orig_generate = talker.code_predictor.generate

    def forced_generate(*a, **k):
        r = orig_generate(*a, **k)
        own_sub.append(r.sequences.clone())
        r.sequences = gold_codes[:, frame_no[0], 1:].clone()
        frame_no[0] += 1
        return r
    talker.code_predictor.generate = forced_generate
    eos = t.codec_eos_token_id
    suppress = [i for i in range(t.vocab_size - 1024, t.vocab_size) if i != eos]
    own_tok, logits_sel = [], {}
    t1 = time.time()
    with torch.no_grad():
        talker.rope_deltas = None
        o = talker(inputs_embeds=emb, attention_mask=mask, use_cache=True, output_hidden_states=True,
                   trailing_text_hidden=trailing, tts_pad_embed=pad)
        T = emb.shape[1]
        for step in range(F + 1):
            raw = o.logits[:, -1].float().clone()
            if step in LOGIT_STEPS:
                logits_sel[step] = raw.clone()
            s = raw.clone()
            if step > 0:                              # HF processors on the GOLDEN history (teacher forcing)
                hist = gold_tokens[:, :step]
                sc = torch.gather(s, 1, hist)
                sc = torch.where(sc < 0, sc * 1.05, sc / 1.05)
                s = s.scatter(1, hist, sc)
            s[:, eos] = float("-inf")                 # min_new_tokens = max_new_tokens in this configuration
            s[:, suppress] = float("-inf")
            own_tok.append(torch.argmax(s, dim=-1))
            if step == F:
                break
            tok = gold_tokens[:, step]
            mask = torch.cat([mask, mask.new_ones(B, 1)], 1)
            o = talker(input_ids=tok[:, None], attention_mask=mask, past_key_values=o.past_key_values, use_cache=True,
                       cache_position=torch.tensor([T + step]), past_hidden=o.past_hidden,
                       generation_step=o.generation_step, trailing_text_hidden=trailing, tts_pad_embed=pad,
                       output_hidden_states=True, subtalker_dosample=False, subtalker_top_k=None,
                       subtalker_top_p=None, subtalker_temperature=None)
            if step % 10 == 0:
                print(f"  bf16 teacher-forced step {step}/{F} ({time.time() - t1:.0f}s)", flush=True)
    dt = time.time() - t1
    own_tokens = torch.stack(own_tok, 1)              # (B, F + 1): the bf16 reference's choice where the golden has tokens[:, i]
    own = torch.stack(own_sub, 1)                     # (B, F, 15)
    agree0 = float((own_tokens == gold_tokens).float().mean())
    agree_sub = float((own == gold_codes[:, :, 1:]).float().mean())
    np.savez_compressed(os.path.join(GOLDEN, "talker_17b_b8_bf16.npz"), own_tokens=own_tokens.numpy(), own_sub=own.numpy(),
                        logit_steps=np.array(sorted(logits_sel)), logits_sel=torch.stack([logits_sel[k] for k in sorted(logits_sel)], 0).numpy(),
                        agree_cb0=agree0, agree_sub=agree_sub, ref_cpu_seconds=dt, ref_cpu_threads=torch.get_num_threads())
    print(f"talker_17b_b8_bf16: reference-in-bf16 vs reference-in-fp32, teacher-forced: cb-0 top-1 agreement {agree0:.4f}, "
          f"sub-codebook agreement {agree_sub:.4f}, {dt:.0f}s")


def gen_prompt_tiny():
    """Pin the prompt assembly (M:2068-2269) by recording what generate() hands to talker.generate."""
    import torch
    t = synth.talker_tiny()
    w = synth.talker_weights(t)
    TalkerConfig, TopConfig, tk = ref_talker_cfgs(t)
    from qwen_tts.core.models.modeling_qwen3_tts import Qwen3TTSForConditionalGeneration
    top = TopConfig(talker_config=tk, tts_model_type="custom_voice", tts_model_size="tiny", tokenizer_type="12hz",
                    im_start_token_id=t.im_start_token_id, im_end_token_id=t.im_end_token_id,
                    tts_pad_token_id=t.tts_pad_token_id, tts_bos_token_id=t.tts_bos_token_id,
                    tts_eos_token_id=t.tts_eos_token_id)
    top.talker_config._attn_implementation = "eager"
    top.talker_config.code_predictor_config._attn_implementation = "eager"
    model = Qwen3TTSForConditionalGeneration(top).eval()
    _load(model.talker, w)
    g = np.random.default_rng(31)
    rec = {}

    class Stop(Exception):
        pass

    def recorder(**kw):
        rec.update(kw)
        raise Stop()
    model.talker.generate = recorder
    out = {"weights_checksum": synth.weights_checksum(w)}
    a, n = 77, 198
    cases = {
        "cv_ns": dict(non_streaming_mode=True, speakers=["vivian", "ryan", "vivian"], languages=["chinese", "english", "auto"], instruct=[None, 6, None]),
        "cv_st": dict(non_streaming_mode=False, speakers=["vivian", "ryan", "vivian"], languages=["chinese", "english", "auto"], instruct=[None, 6, None]),
        "vd_st": dict(non_streaming_mode=False, speakers=None, languages=["auto", "english"], instruct=[5, 9]),
    }
    # voice-clone prompts (Base model, M:2188-2197, 1968-2019): precomputed ref_code / x-vector, ICL + x-vector-only rows
    vc = {
        "vc_st": dict(non_streaming_mode=False, speakers=None, languages=["english", "auto", "chinese"], instruct=[None, None, None],
                      icl=[True, False, True], ref_frames=[7, 0, 30]),
        "vc_ns": dict(non_streaming_mode=True, speakers=None, languages=["english", "auto", "chinese"], instruct=[None, None, None],
                      icl=[True, False, True], ref_frames=[7, 0, 30]),
    }
    cases.update(vc)
    for cname, cs in cases.items():
        B = len(cs["languages"])
        ids, ins = [], []
        for i in range(B):
            body = g.integers(0, 490, (int(g.integers(6, 14)),)).tolist()
            ids.append(torch.tensor([[t.im_start_token_id, a, n] + body + [t.im_end_token_id, n, t.im_start_token_id, a, n]]))
            k = cs["instruct"][i]
            ins.append(None if k is None else torch.tensor([[t.im_start_token_id] + g.integers(0, 490, (k,)).tolist() + [t.im_end_token_id, n]]))
        extra = {}
        if "icl" in cs:
            ref_ids, ref_code, spk = [], [], []
            for i in range(B):
                rbody = g.integers(0, 490, (int(g.integers(3, 9)),)).tolist()
                ref_ids.append(torch.tensor([[t.im_start_token_id, a, n] + rbody + [t.im_end_token_id, n]]))
                nf = cs["ref_frames"][i]
                rc = np.concatenate([g.integers(0, 256, (nf, 1)), g.integers(0, t.cp_vocab_size, (nf, t.num_code_groups - 1))], 1) if nf else None
                ref_code.append(None if rc is None else torch.from_numpy(rc))
                spk.append(torch.from_numpy(g.standard_normal(t.hidden_size).astype(np.float32) * 0.1))
                out[f"{cname}_refids{i}"] = ref_ids[i].numpy()
                if rc is not None:
                    out[f"{cname}_refcode{i}"] = rc
                out[f"{cname}_spk{i}"] = spk[i].numpy()
            extra = dict(ref_ids=ref_ids, voice_clone_prompt=dict(ref_code=ref_code, ref_spk_embedding=spk,
                                                                  x_vector_only_mode=[not x for x in cs["icl"]], icl_mode=cs["icl"]))
        try:
            model.generate(input_ids=ids, instruct_ids=ins, languages=cs["languages"], speakers=cs["speakers"],
                           non_streaming_mode=cs["non_streaming_mode"], do_sample=False, subtalker_dosample=False, **extra)
        except Stop:
            pass
        for i in range(B):
            out[f"{cname}_ids{i}"] = ids[i].numpy()
            if ins[i] is not None:
                out[f"{cname}_ins{i}"] = ins[i].numpy()
        out[f"{cname}_embeds"] = rec["inputs_embeds"].detach().numpy()
        out[f"{cname}_mask"] = rec["attention_mask"].numpy()
        out[f"{cname}_trailing"] = rec["trailing_text_hidden"].detach().numpy()
        out[f"{cname}_tts_pad"] = rec["tts_pad_embed"].detach().numpy()
        out[f"{cname}_suppress"] = np.array(rec["suppress_tokens"])
        out[f"{cname}_eos"] = rec["eos_token_id"]
        out[f"{cname}_min_new"] = rec["min_new_tokens"]
        print(cname, rec["inputs_embeds"].shape, rec["trailing_text_hidden"].shape)
    np.savez_compressed(os.path.join(GOLDEN, "prompt_tiny.npz"), **out)


def gen_ckpt_tiny():
    """The JSON side of a checkpoint directory, written by the REFERENCE's own config classes, so that the mirrored
    `from_pretrained` is tested against the reference's serialisation (nested talker_config / code_predictor_config /
    decoder_config, rope_scaling, speaker tables...).  Weights and the text tokenizer are produced by the test itself
    (tests/ckpt_util.py) from synth.py."""
    ref_shims.install()
    import json
    from qwen_tts.core.tokenizer_12hz.configuration_qwen3_tts_tokenizer_v2 import (Qwen3TTSTokenizerV2Config,
                                                                                   Qwen3TTSTokenizerV2DecoderConfig)
    t = synth.talker_tiny()
    c = synth.codec_tiny()
    c.codebook_size = t.cp_vocab_size                 # the codec codebooks must cover the talker's code range
    _, TopConfig, tk = ref_talker_cfgs(t)
    top = TopConfig(talker_config=tk, tts_model_type="custom_voice", tts_model_size="tiny", tokenizer_type="12hz",
                    im_start_token_id=t.im_start_token_id, im_end_token_id=t.im_end_token_id,
                    tts_pad_token_id=t.tts_pad_token_id, tts_bos_token_id=t.tts_bos_token_id,
                    tts_eos_token_id=t.tts_eos_token_id)
    dec = Qwen3TTSTokenizerV2DecoderConfig(
        codebook_size=c.codebook_size, codebook_dim=c.codebook_dim, hidden_size=c.hidden_size,
        latent_dim=c.latent_dim, num_attention_heads=c.num_attention_heads,
        num_key_value_heads=c.num_key_value_heads, head_dim=c.head_dim, sliding_window=c.sliding_window,
        intermediate_size=c.intermediate_size, num_hidden_layers=c.num_hidden_layers,
        num_quantizers=c.num_quantizers, upsample_rates=c.upsample_rates,
        upsampling_ratios=c.upsampling_ratios, decoder_dim=c.decoder_dim, rms_norm_eps=c.rms_norm_eps,
        rope_theta=c.rope_theta, max_position_embeddings=c.max_position_embeddings)
    tokc = Qwen3TTSTokenizerV2Config(decoder_config=dec.to_dict(), decode_upsample_rate=c.total_upsample,
                                     encode_downsample_rate=c.total_upsample)
    out = os.path.join(GOLDEN, "ckpt_tiny")
    os.makedirs(os.path.join(out, "speech_tokenizer"), exist_ok=True)
    with open(os.path.join(out, "config.json"), "w") as f:
        f.write(top.to_json_string())
    with open(os.path.join(out, "speech_tokenizer", "config.json"), "w") as f:
        f.write(tokc.to_json_string())
    with open(os.path.join(out, "generation_config.json"), "w") as f:      # the keys the wrapper reads (IM:287-352)
        json.dump(dict(do_sample=True, top_k=50, top_p=1.0, temperature=0.9, repetition_penalty=1.05,
                       subtalker_dosample=True, subtalker_top_k=50, subtalker_top_p=1.0, subtalker_temperature=0.9,
                       max_new_tokens=8192), f, indent=1)
    print("ckpt_tiny: wrote", sorted(os.listdir(out)))


def gen_speaker_tiny():
    """Pin oracle/speaker_ref.py to the reference's Qwen3TTSSpeakerEncoder (M:95-393) and to the
    arithmetic of its mel_spectrogram (M:402-464) fed with the restated Slaney filterbank (librosa is absent here)."""
    ref_shims.install()
    import torch
    import speaker_ref
    from qwen_tts.core.models import modeling_qwen3_tts as M
    from qwen_tts.core.models.configuration_qwen3_tts import Qwen3TTSSpeakerEncoderConfig
    c = synth.speaker_tiny()
    w = synth.speaker_weights(c)
    cfg = Qwen3TTSSpeakerEncoderConfig(mel_dim=c.mel_dim, enc_dim=c.enc_dim, enc_channels=list(c.enc_channels),
                                       enc_kernel_sizes=list(c.enc_kernel_sizes), enc_dilations=list(c.enc_dilations),
                                       enc_attention_channels=c.enc_attention_channels, enc_res2net_scale=c.enc_res2net_scale,
                                       enc_se_channels=c.enc_se_channels)
    m = M.Qwen3TTSSpeakerEncoder(cfg).eval()
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == synth.speaker_param_shapes(c)
    _load(m, w)
    g = np.random.default_rng(31)
    mels = torch.from_numpy(g.standard_normal((2, 41, c.mel_dim)).astype(np.float32))
    audio = (g.standard_normal(12000) * 0.2).clip(-1, 1).astype(np.float32)
    fb = speaker_ref.mel_filterbank_slaney(24000, 1024, 128, 0, 12000)
    M.librosa_mel_fn = lambda sr, n_fft, n_mels, fmin, fmax: fb                 # the stubbed third-party call
    with torch.no_grad():
        emb = m(mels)
        mel = M.mel_spectrogram(torch.from_numpy(audio).unsqueeze(0), n_fft=1024, num_mels=128, sampling_rate=24000,
                                hop_size=256, win_size=1024, fmin=0, fmax=12000)
    np.savez_compressed(os.path.join(GOLDEN, "speaker_tiny.npz"), weights_checksum=synth.weights_checksum(w),
                        mels=mels.numpy(), embedding=emb.numpy(), audio=audio, mel=mel.numpy(),
                        fb_checksum=float(np.abs(fb).sum()), fb_row0=fb[0, :8], fb_peak_bins=fb.argmax(1).astype(np.int32))
    print("speaker_tiny: embedding", tuple(emb.shape), "mel", tuple(mel.shape))


def ref_mimi_encoder(c, w):
    """The reference's encoder class (tokenizer v2:897-908) on a MimiConfig built from synth.MimiEncCfg."""
    ref_shims.install()
    from transformers import MimiConfig
    from qwen_tts.core.tokenizer_12hz.modeling_qwen3_tts_tokenizer_v2 import Qwen3TTSTokenizerV2Encoder
    hop = int(np.prod(c.upsampling_ratios))
    mc = MimiConfig(sampling_rate=c.sampling_rate, hidden_size=c.hidden_size, num_filters=c.num_filters,
                    num_residual_layers=c.num_residual_layers, upsampling_ratios=list(c.upsampling_ratios),
                    kernel_size=c.kernel_size, last_kernel_size=c.last_kernel_size, residual_kernel_size=c.residual_kernel_size,
                    dilation_growth_rate=c.dilation_growth_rate, compress=c.compress, codebook_size=c.codebook_size,
                    codebook_dim=c.codebook_dim, vector_quantization_hidden_dimension=c.codebook_dim,
                    num_quantizers=c.num_quantizers, num_semantic_quantizers=c.num_semantic_quantizers,
                    num_hidden_layers=c.num_hidden_layers, intermediate_size=c.intermediate_size,
                    num_attention_heads=c.num_attention_heads, num_key_value_heads=c.num_key_value_heads, head_dim=c.head_dim,
                    sliding_window=c.sliding_window, norm_eps=c.norm_eps, upsample_groups=c.hidden_size,
                    frame_rate=c.sampling_rate / (hop * 2))
    mc._attn_implementation = "eager"
    m = Qwen3TTSTokenizerV2Encoder(mc).eval()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == synth.mimi_enc_param_shapes(c)
    _load(m, w)
    for mod in m.modules():                  # the codebook caches embed_sum / cluster_usage lazily
        if hasattr(mod, "_embed"):
            mod._embed = None
    return m


def gen_codec_enc_small():
    """The same pin at GEMM-friendly dimensions (synth.mimi_enc_small): the fixture the HIP encoder is tested against."""
    import torch
    c = synth.mimi_enc_small()
    w = synth.mimi_enc_weights(c)
    m = ref_mimi_encoder(c, w)
    g = np.random.default_rng(43)
    out = {"weights_checksum": synth.weights_checksum(w)}
    for n in (16, 203, 331):
        x = torch.from_numpy((g.standard_normal((2, 1, n)) * 0.5).astype(np.float32))
        with torch.no_grad():
            out[f"wav{n}"] = x.numpy()
            out[f"codes{n}"] = m.encode(input_values=x, return_dict=True).audio_codes.numpy()
    np.savez_compressed(os.path.join(GOLDEN, "codec_enc_small.npz"), **out)
    print("codec_enc_small:", {k: v.shape for k, v in out.items() if k.startswith("codes")})


def gen_codec_enc_tiny():
    """Pin oracle/codec_enc_ref.py: MimiModel.encode through the reference's encoder class, and the body
    of Qwen3TTSTokenizerV2Model.encode (v2:961-991: first 4 codebooks here, per-row trim from the padding mask) called
    unbound on a stand-in `self`."""
    import torch
    ref_shims.install()
    from qwen_tts.core.tokenizer_12hz.modeling_qwen3_tts_tokenizer_v2 import Qwen3TTSTokenizerV2Model
    c = synth.mimi_enc_tiny()
    w = synth.mimi_enc_weights(c)
    m = ref_mimi_encoder(c, w)
    g = np.random.default_rng(41)
    out = {"weights_checksum": synth.weights_checksum(w)}
    for n in (16, 160, 203, 331):
        x = torch.from_numpy((g.standard_normal((2, 1, n)) * 0.5).astype(np.float32))
        with torch.no_grad():
            out[f"wav{n}"] = x.numpy()
            out[f"codes{n}"] = m.encode(input_values=x, return_dict=True).audio_codes.numpy()
    lens = [331, 170]
    wav = np.zeros((2, 331), np.float32)
    mask = np.zeros((2, 331), np.int64)
    for i, l in enumerate(lens):
        wav[i, :l] = (g.standard_normal(l) * 0.5).astype(np.float32)
        mask[i, :l] = 1
    fake = types.SimpleNamespace(config=types.SimpleNamespace(return_dict=True), encoder=m,
                                 encoder_valid_num_quantizers=c.encoder_valid_num_quantizers,
                                 encode_downsample_rate=c.encode_downsample_rate)
    with torch.no_grad():
        codes = Qwen3TTSTokenizerV2Model.encode(fake, torch.from_numpy(wav), torch.from_numpy(mask)).audio_codes
    out.update(batch_wav=wav, batch_mask=mask)
    for i, cd in enumerate(codes):
        out[f"batch_codes{i}"] = cd.numpy()
    np.savez_compressed(os.path.join(GOLDEN, "codec_enc_tiny.npz"), **out)
    print("codec_enc_tiny:", {k: v.shape for k, v in out.items() if k.startswith(("codes", "batch_codes"))})


def gen_codec_enc_real():
    """f3 at the RELEASED dimensions: the reference's encoder class (tokenizer v2:897-908; Mimi: hidden 512, 8 layers,
    32 codebooks of 2048 x 256) on 3 s of audio (72 000 samples), batch 2.  The waveform is synth.rand_audio(seed) (not stored); stored are
    the reference's codes, and -- from the oracle restatement, which must reproduce those codes here -- the relative gap between the nearest
    and the second-nearest codebook entry of every index (the near-tie exemption of the GPU test)."""
    import torch
    import codec_enc_ref
    c = synth.mimi_enc_real()
    w = synth.mimi_enc_weights(c)
    m = ref_mimi_encoder(c, w)
    seed, n = 77, 72000
    x = torch.from_numpy(synth.rand_audio(seed, 2, n))[:, None]
    with torch.no_grad():
        codes = m.encode(input_values=x, return_dict=True).audio_codes.numpy()
        mg = []
        oc = codec_enc_ref.mimi_encode({k: torch.from_numpy(v) for k, v in w.items()}, c, x, margins=mg).numpy()
    margin = torch.stack(mg, 1).numpy().astype(np.float32)          # (B, Q, T)
    agree = float((oc == codes).mean())
    print(f"codec_enc_real: codes {codes.shape}, oracle == reference on {agree:.4f} of the indices, smallest margin {margin.min():.2e}")
    assert agree >= 0.999
    np.savez_compressed(os.path.join(GOLDEN, "codec_enc_real.npz"), weights_checksum=synth.weights_checksum(w), seed=seed, samples=n,
                        codes=codes.astype(np.int16), margin=margin, oracle_agree=agree)


def gen_speaker_real():
    """f4 at the RELEASED dimensions: the reference's Qwen3TTSSpeakerEncoder (M:95-393; ECAPA-TDNN 512/512/512/512/1536, enc_dim 2048)
    and its mel_spectrogram (M:402-464, with the restated Slaney filterbank as in speaker_tiny) on 3 s of audio, batch 2."""
    ref_shims.install()
    import torch
    import speaker_ref
    from qwen_tts.core.models import modeling_qwen3_tts as M
    from qwen_tts.core.models.configuration_qwen3_tts import Qwen3TTSSpeakerEncoderConfig
    c = synth.speaker_real()
    w = synth.speaker_weights(c)
    cfg = Qwen3TTSSpeakerEncoderConfig(mel_dim=c.mel_dim, enc_dim=c.enc_dim, enc_channels=list(c.enc_channels),
                                       enc_kernel_sizes=list(c.enc_kernel_sizes), enc_dilations=list(c.enc_dilations),
                                       enc_attention_channels=c.enc_attention_channels, enc_res2net_scale=c.enc_res2net_scale,
                                       enc_se_channels=c.enc_se_channels)
    m = M.Qwen3TTSSpeakerEncoder(cfg).eval()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == synth.speaker_param_shapes(c)
    _load(m, w)
    seed, n = 78, 72000
    audio = torch.from_numpy(synth.rand_audio(seed, 2, n))
    fb = speaker_ref.mel_filterbank_slaney(24000, 1024, 128, 0, 12000)
    M.librosa_mel_fn = lambda sr, n_fft, n_mels, fmin, fmax: fb
    with torch.no_grad():
        mel = M.mel_spectrogram(audio, n_fft=1024, num_mels=128, sampling_rate=24000, hop_size=256, win_size=1024, fmin=0, fmax=12000)
        emb = m(mel.transpose(1, 2))                               # M:1951: mels (B, frames, 128)
        oemb = speaker_ref.speaker_encoder_forward({k: torch.from_numpy(v) for k, v in w.items()}, c,
                                                   speaker_ref.mel_spectrogram(audio).transpose(1, 2))
    d = float((emb - oemb).abs().max())
    print(f"speaker_real: embedding {tuple(emb.shape)} |max| {float(emb.abs().max()):.3f}, mel {tuple(mel.shape)}, oracle - reference max {d:.2e}")
    assert d <= 1e-4 * max(1.0, float(emb.abs().max()))
    np.savez_compressed(os.path.join(GOLDEN, "speaker_real.npz"), weights_checksum=synth.weights_checksum(w), seed=seed, samples=n,
                        embedding=emb.numpy(), mel_sum=float(mel.double().sum()), mel_frames=mel.shape[-1])


def gen_codec_real_bf16():
    """The reference's OWN decoder run in **bfloat16** (V2:869-896; the dtype of its examples) on codec_real.npz's 10 s of codes:
    the yardstick for the MI355X bf16 codec engine -- how far does bf16 move the reference's own waveform from its fp32 waveform,
    and is the engine inside that distance?  Stored: the bf16 waveform (float16 is enough for a yardstick) and its relative RMS
    distance to the fp32 golden."""
    import torch
    c = synth.codec_real()
    w = synth.codec_weights(c)
    g = np.load(os.path.join(GOLDEN, "codec_real.npz"))
    assert abs(synth.weights_checksum(w) - float(g["weights_checksum"])) < 1e-3
    dec = ref_codec_decoder(c, w).to(torch.bfloat16)
    for mod in dec.modules():             # rotary inv_freq is a non-persistent fp32 buffer in the reference: keep it fp32
        if hasattr(mod, "rope_init_fn") and hasattr(mod, "inv_freq"):
            inv, _ = mod.rope_init_fn(mod.config, "cpu")
            mod.inv_freq = inv
            mod.original_inv_freq = inv
    codes = torch.from_numpy(g["t125_codes"].astype(np.int64))
    t = time.time()
    with torch.no_grad():
        wav = ref_model_decode(dec, c, codes)[0].float().numpy()
    dt = time.time() - t
    ref = g["t125_wav"].astype(np.float64)
    rel = float(np.sqrt(((wav - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()))
    np.savez_compressed(os.path.join(GOLDEN, "codec_real_bf16.npz"), t125_wav_bf16=wav.astype(np.float16), rel_rms_vs_fp32=rel,
                        ref_cpu_seconds=dt, weights_checksum=float(g["weights_checksum"]))
    print(f"codec_real_bf16: reference-in-bf16 vs reference-in-fp32 waveform, relative RMS {rel:.4f} ({dt:.0f}s)")


ICL_TEXT_LENS = [20, 34, 12, 27, 41, 16, 30, 23]       # body ids of the 8 requests
ICL_REF_TEXT = [16, 12, 16, 20, 16, 9, 16, 14]         # ref-text ids
ICL_REF_FRAMES = [38, 30, 38, 45, 38, 25, 52, 38]      # ref_code frames (3 s of reference audio = 38 frames)


def gen_talker_17b_base_icl_b8():
    """BASELINE config 5's request shape at REAL dims: Qwen3-TTS-12Hz-1.7B **Base** (voice clone), ICL prompt = ref text ids +
    `ref_code` (38 x 16 and ragged neighbours) + x-vector speaker embedding (M:1968-2019 `generate_icl_prompt`, M:2188-2197;
    wrapper IM:470-631), batch 8, streaming text input (the wrapper's default), greedy, 48 frames.  The reference's own
    `Qwen3TTSForConditionalGeneration.generate` assembles the prompt (full 151 936-row text embedding, text_projection, the
    16-codebook embedding sum of every reference frame) and hands it to `talker.generate`, which is the restated HF-4.57.3 loop
    around the reference's talker.forward as in every other talker fixture.  Stored: the request (ids / ref ids / ref codes /
    x-vectors are regenerated from the seed by the test through `icl_requests`), a strided sample + per-row sums of the assembled
    embeddings, and the greedy codes."""
    import torch
    t = synth.talker_17b()
    w = synth.talker_weights(t, with_text=True)
    TalkerConfig, TopConfig, tk = ref_talker_cfgs(t)
    from qwen_tts.core.models.modeling_qwen3_tts import Qwen3TTSForConditionalGeneration
    top = TopConfig(talker_config=tk, tts_model_type="base", tts_model_size="1b7", tokenizer_type="12hz",
                    im_start_token_id=t.im_start_token_id, im_end_token_id=t.im_end_token_id,
                    tts_pad_token_id=t.tts_pad_token_id, tts_bos_token_id=t.tts_bos_token_id,
                    tts_eos_token_id=t.tts_eos_token_id)
    top.talker_config._attn_implementation = "eager"
    top.talker_config.code_predictor_config._attn_implementation = "eager"
    with torch.device("meta"):
        model = Qwen3TTSForConditionalGeneration(top)
    model = model.to_empty(device="cpu").eval()
    for mod in model.modules():
        if hasattr(mod, "rope_init_fn") and hasattr(mod, "inv_freq"):
            inv, _ = mod.rope_init_fn(mod.config, "cpu")
            mod.inv_freq = inv
            mod.original_inv_freq = inv
    _load(model.talker, w)
    model.talker.config._attn_implementation = "eager"
    req = synth.icl_requests(t, 200, ICL_TEXT_LENS, ICL_REF_TEXT, ICL_REF_FRAMES)
    rec = {}
    MAX_NEW = 49

    def talker_generate(inputs_embeds=None, attention_mask=None, trailing_text_hidden=None, tts_pad_embed=None, **kw):
        rec.update(embeds=inputs_embeds, mask=attention_mask, trailing=trailing_text_hidden, tts_pad=tts_pad_embed, kw=kw)
        tr = {}
        codes, toks, hidden = restated_sample_loop(model.talker, t, inputs_embeds, attention_mask, trailing_text_hidden,
                                                   tts_pad_embed, max_new_tokens=MAX_NEW, min_new_tokens=kw["min_new_tokens"],
                                                   eos_token_id=kw["eos_token_id"], repetition_penalty=kw["repetition_penalty"],
                                                   trace=tr)
        rec.update(codes=codes, tokens=toks, hidden=hidden, margin=torch.stack(tr["margin"], 1))
        raise StopIteration
    model.talker.generate = talker_generate
    t1 = time.time()
    try:
        model.generate(input_ids=req["ids"], instruct_ids=[None] * 8, ref_ids=req["ref_ids"], voice_clone_prompt=req["vcp"],
                       languages=req["languages"], speakers=None, non_streaming_mode=False, max_new_tokens=MAX_NEW,
                       do_sample=False, subtalker_dosample=False)
    except StopIteration:
        pass
    dt = time.time() - t1
    e = rec["embeds"].numpy()
    out = {"weights_checksum": synth.weights_checksum(w),
           "seed": 200, "text_lens": np.array(ICL_TEXT_LENS), "ref_text": np.array(ICL_REF_TEXT), "ref_frames": np.array(ICL_REF_FRAMES),
           "max_new": MAX_NEW, "min_new": rec["kw"]["min_new_tokens"], "mask": rec["mask"].numpy(),
           "embeds_strided": e[:, :, ::64].copy(), "embeds_rowsum": e.astype(np.float64).sum(-1),
           "trailing_strided": rec["trailing"].numpy()[:, :, ::64].copy(), "trailing_rowsum": rec["trailing"].numpy().astype(np.float64).sum(-1),
           "tts_pad": rec["tts_pad"].numpy(), "codes": rec["codes"].numpy(), "tokens": rec["tokens"].numpy(),
           "margin": rec["margin"].numpy(), "hidden_last": rec["hidden"][:, -1].numpy(), "ref_cpu_seconds": dt}
    np.savez_compressed(os.path.join(GOLDEN, "talker_17b_base_icl_b8.npz"), **out)
    print(f"talker_17b_base_icl_b8: prompt {tuple(e.shape)} trailing {tuple(rec['trailing'].shape)} codes {tuple(rec['codes'].shape)} "
          f"min margin {float(rec['margin'].min()):.5f} ({dt:.0f}s)")


ALL = {"codec_tiny": gen_codec_tiny, "codec_real": gen_codec_real, "talker_tiny": gen_talker_tiny,
       "talker_06b": gen_talker_06b, "talker_17b": gen_talker_17b, "prompt_tiny": gen_prompt_tiny,
       "talker_06b_b8": gen_talker_06b_b8, "talker_17b_b32": gen_talker_17b_b32, "talker_17b_b8": gen_talker_17b_b8, "talker_06b_long": gen_talker_06b_long,
       "talker_17b_b8_bf16": gen_talker_17b_b8_bf16, "ckpt_tiny": gen_ckpt_tiny, "speaker_tiny": gen_speaker_tiny, "codec_enc_tiny": gen_codec_enc_tiny, "codec_enc_small": gen_codec_enc_small,
       "codec_real_bf16": gen_codec_real_bf16, "talker_17b_base_icl_b8": gen_talker_17b_base_icl_b8,
       "codec_enc_real": gen_codec_enc_real, "speaker_real": gen_speaker_real}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=",".join(ALL))
    args = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    import torch
    torch.set_num_threads(os.cpu_count())
    for k in args.only.split(","):
        t = time.time()
        ALL[k]()
        print(f"[{k}] done in {time.time() - t:.1f}s")
