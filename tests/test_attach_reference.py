"""CPU, build container only (needs /root/reference): INTEGRATION.md §B executed.

`qwen3_tts_amd.attach(model)` patches a loaded REFERENCE `Qwen3TTSForConditionalGeneration` at seam S2
(`talker.generate`, modeling_qwen3_tts.py:2272) and seam S4 (`speech_tokenizer.model.decode`, tokenizer v2:993).  Here the
reference's own classes are instantiated with tiny synthetic weights (oracle/ref_shims.py bridges the transformers 4.57 ->
5.x API drift, nothing else), the engines behind `attach` are the host-emulation build of the product's C++ / HIP sources
(tests/hostemu), and the reference's OWN `generate()` -- prompt assembly M:2068-2269, EOS trim M:2280-2292 -- runs unmodified
around the patched seam.  Baseline for the comparison: the same reference `generate()` with `talker.generate` served by a
greedy HF-4.57.3 `_sample` loop around the reference's own `talker.forward` that hands back the forward's real
`hidden_states` exactly as HF collects them (the installed transformers 5.x loop can no longer drive this forward --
SURVEY.md 8c -- which is why the loop is restated here too).  On the GPU box /root/reference does not exist: skipped there."""
import os
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("QTTS_REFERENCE_ROOT", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "qwen_tts")), reason="reference tree not present")


@pytest.fixture(scope="module")
def emu():
    sys.path.insert(0, os.path.join(HERE, "hostemu"))
    import pyshim
    pyshim.install()
    try:
        yield
    finally:
        pyshim.uninstall()


def _hf_like_generate(talker, t):
    """Greedy HF-4.57.3 `_sample` around the REFERENCE talker.forward, returning what `generate(return_dict_in_generate=True,
    output_hidden_states=True)` returns as far as M:2280-2281 reads it: `hidden_states[i]` = the i-th forward's own
    `outputs.hidden_states`."""
    def generate(inputs_embeds=None, attention_mask=None, trailing_text_hidden=None, tts_pad_embed=None, max_new_tokens=None,
                 min_new_tokens=2, eos_token_id=None, repetition_penalty=1.05, suppress_tokens=None, **kw):
        assert not kw.get("do_sample") and not kw.get("subtalker_dosample")
        B, T, _ = inputs_embeds.shape
        mask = attention_mask
        talker.rope_deltas = None
        o = talker(inputs_embeds=inputs_embeds, attention_mask=mask, use_cache=True, output_hidden_states=True,
                   trailing_text_hidden=trailing_text_hidden, tts_pad_embed=tts_pad_embed)
        hs = [o.hidden_states]
        generated = torch.zeros(B, 0, dtype=torch.long)
        unfinished = torch.ones(B, dtype=torch.long)
        step = 0
        while True:
            s = o.logits[:, -1].float().clone()
            if generated.shape[1] > 0:
                sc = torch.gather(s, 1, generated)
                sc = torch.where(sc < 0, sc * repetition_penalty, sc / repetition_penalty)
                s = s.scatter(1, generated, sc)
            if generated.shape[1] < min_new_tokens:
                s[:, eos_token_id] = float("-inf")
            s[:, suppress_tokens] = float("-inf")
            tok = torch.argmax(s, dim=-1)
            tok = tok * unfinished + eos_token_id * (1 - unfinished)
            generated = torch.cat((generated, tok[:, None]), dim=1)
            unfinished = unfinished & (tok != eos_token_id).long()
            if generated.shape[1] >= max_new_tokens or unfinished.max() == 0:
                break
            mask = torch.cat([mask, mask.new_ones(B, 1)], 1)
            o = talker(input_ids=tok[:, None], attention_mask=mask, past_key_values=o.past_key_values, use_cache=True,
                       cache_position=torch.tensor([T + step]), past_hidden=o.past_hidden, generation_step=o.generation_step,
                       trailing_text_hidden=trailing_text_hidden, tts_pad_embed=tts_pad_embed, output_hidden_states=True,
                       subtalker_dosample=False, subtalker_top_k=None, subtalker_top_p=None, subtalker_temperature=None)
            hs.append(o.hidden_states)
            step += 1
        return types.SimpleNamespace(hidden_states=hs, sequences=generated)
    return generate


def test_attach_patches_the_reference_model_at_both_seams(emu):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gen_golden as gg
    import synth
    import qwen3_tts_amd
    t = synth.talker_tiny()
    w = synth.talker_weights(t)
    _, TopConfig, tk = gg.ref_talker_cfgs(t)
    from qwen_tts.core.models.modeling_qwen3_tts import Qwen3TTSForConditionalGeneration
    top = TopConfig(talker_config=tk, tts_model_type="custom_voice", tts_model_size="tiny", tokenizer_type="12hz",
                    im_start_token_id=t.im_start_token_id, im_end_token_id=t.im_end_token_id, tts_pad_token_id=t.tts_pad_token_id,
                    tts_bos_token_id=t.tts_bos_token_id, tts_eos_token_id=t.tts_eos_token_id)
    top.talker_config._attn_implementation = "eager"
    top.talker_config.code_predictor_config._attn_implementation = "eager"
    model = Qwen3TTSForConditionalGeneration(top).eval()
    gg._load(model.talker, w)
    # a reference codec decoder behind the reference model's `speech_tokenizer.model` attribute (seam S4)
    c = synth.codec_tiny()
    c.codebook_size = t.cp_vocab_size
    cw = synth.codec_weights(c)
    ref_dec = gg.ref_codec_decoder(c, cw)
    model.speech_tokenizer = types.SimpleNamespace(model=types.SimpleNamespace(config=synth.cfg_dict(c), decoder=ref_dec,
                                                                               dtype=torch.float32))
    g = np.random.default_rng(3)
    a, n = 77, 198
    ids, ins = [], []
    for i, k in enumerate((9, 13, 6)):
        body = g.integers(0, 490, (k,)).tolist()
        ids.append(torch.tensor([[t.im_start_token_id, a, n] + body + [t.im_end_token_id, n, t.im_start_token_id, a, n]]))
        ins.append(None if i != 1 else torch.tensor([[t.im_start_token_id] + g.integers(0, 490, (5,)).tolist() + [t.im_end_token_id, n]]))
    kw = dict(input_ids=ids, instruct_ids=ins, languages=["chinese", "english", "auto"], speakers=["vivian", "ryan", "vivian"],
              non_streaming_mode=False, do_sample=False, subtalker_dosample=False, max_new_tokens=9)
    # ---- baseline: the reference's generate() around the reference's own forward
    model.talker.generate = _hf_like_generate(model.talker, t)
    with torch.no_grad():
        ref_codes, ref_hidden = model.generate(**kw)
    # ---- patched: INTEGRATION.md §B
    qwen3_tts_amd.attach(model, max_batch=4, max_seq=128, talker_device="cpu")
    with torch.no_grad():
        codes, hidden = model.generate(**kw)
    assert len(codes) == len(ref_codes) == 3
    for x, y in zip(codes, ref_codes):
        assert x.shape == y.shape and torch.equal(x.cpu(), y), "codes through the patched seam differ from the reference's"
    for x, y in zip(hidden, ref_hidden):
        assert x.shape == y.shape and float((x.cpu() - y).abs().max()) <= 2e-4
    # ---- seam S4: decode of the padded code batch, as Qwen3TTSTokenizer.decode calls it (IT:259 -> v2:993)
    padded = torch.nn.utils.rnn.pad_sequence(ref_codes, batch_first=True, padding_value=-1)
    out = model.speech_tokenizer.model.decode(padded)
    with torch.no_grad():
        want = gg.ref_model_decode(ref_dec, c, padded)
    assert len(out.audio_values) == len(want)
    for x, y in zip(out.audio_values, want):
        assert x.shape == y.shape and float(torch.sqrt(((x.cpu() - y) ** 2).mean())) <= 1e-4
