"""INTEGRATION.md §B as code: patch a loaded *reference* model in place at its two inner seams.

A maintainer keeps the reference's loaders / text tokenizer / wrappers (`qwen_tts.Qwen3TTSModel`) and swaps only
  * seam S2: `model.talker.generate(...)` (modeling_qwen3_tts.py:2272-2278 -- HF `_sample` around M:1636-1744) and
  * seam S4: `model.speech_tokenizer.model.decode(...)` (modeling_qwen3_tts_tokenizer_v2.py:993-1024)
for the HIP engines.  Everything the reference does around those calls -- prompt assembly (M:2068-2269), EOS trim
(M:2280-2292), wrapper post-processing -- keeps running unchanged, so this file must hand back exactly the structures that
code reads.  Executed in the CPU suite against the reference's own classes (tests/test_attach_reference.py).

    # qwen_tts/inference/qwen3_tts_model.py:118, one added line:
    #     model = qwen3_tts_amd.attach(model) if model.device.type == "cuda" else model
"""
from typing import Any, Optional

import torch

from .codec import CodecDecoderEngine
from .talker import TalkerEngine


class _TalkerGenerateOutput:
    """What M:2280-2281 reads back from HF's generate output: `hidden_states` has one entry per forward =
    (tuple of layer hiddens, codec_ids of the frame that forward consumed).  Entry 0 is the prefill (codec_ids None),
    entry i + 1 is the decode forward that consumed frame i; M:2281 takes `hid[0][-1][:, -1:]` of every entry and
    drops the last one, i.e. it wants `past_hidden` of frame i at position i."""

    def __init__(self, o):
        n = o.n_frames
        hs = []
        for j in range(n + 1):
            k = min(j, n - 1) if n > 0 else 0
            h = o.hidden[:, k:k + 1] if n > 0 else o.hidden[:, :0]
            hs.append(((h,), None if j == 0 else o.codes[:, j - 1]))
        self.hidden_states = hs
        self.sequences = o.tokens


def attach(model: Any, max_batch: int = 8, max_seq: int = 4096, use_graph: bool = True,
           talker_device: Optional[str] = None):
    """model: the reference's `Qwen3TTSForConditionalGeneration`, already loaded.  Returns the same object with its two
    hot seams served by the MI355X engines (weights are taken from the live modules' `state_dict()`)."""
    talker = model.talker
    dev = talker_device or str(model.device)
    wdtype = next(talker.parameters()).dtype
    eng = TalkerEngine(model.config, dict(talker.state_dict()), weight_dtype=wdtype if wdtype == torch.bfloat16 else torch.float32,
                       device=dev, max_batch=max_batch, max_seq=max_seq, use_graph=use_graph)

    def generate(inputs_embeds=None, attention_mask=None, trailing_text_hidden=None, tts_pad_embed=None, **kw):
        kw.pop("output_hidden_states", None)
        kw.pop("return_dict_in_generate", None)
        return _TalkerGenerateOutput(eng.generate(inputs_embeds, attention_mask, trailing_text_hidden, tts_pad_embed,
                                                  output_hidden_states=True, **kw))
    talker.generate = generate                                        # seam S2
    model._mi355x_talker = eng

    st = getattr(getattr(model, "speech_tokenizer", None), "model", None)      # Qwen3TTSTokenizerV2Model, when one is loaded
    if st is not None:
        sdtype = next(st.decoder.parameters()).dtype
        dec = CodecDecoderEngine(st.config, dict(st.decoder.state_dict()),
                                 compute_dtype=sdtype if sdtype == torch.bfloat16 else torch.float32, device=dev, max_batch=max_batch)

        def decode(audio_codes, return_dict=None):
            out = []
            for b0 in range(0, audio_codes.shape[0], dec.max_batch):
                wav, lens = dec.decode_padded(audio_codes[b0:b0 + dec.max_batch])
                out += [w[:l] for w, l in zip(wav, lens)]
            if return_dict is False:
                return (out,)
            return type("Qwen3TTSTokenizerV2DecoderOutput", (), {"audio_values": out})()
        st.decode = decode                                            # seam S4
        model._mi355x_codec = dec
    return model
