"""Build a complete synthetic checkpoint directory in the layout `Qwen3TTSModel.from_pretrained` reads
(qwen_tts/inference/qwen3_tts_model.py:82-121, modeling_qwen3_tts.py:1886-1938):

    config.json, generation_config.json            <- tests/golden/ckpt_tiny (written by the REFERENCE's config classes)
    model.safetensors                               <- synth.py talker weights under the `talker.` prefix
    tokenizer.json, tokenizer_config.json           <- a tiny character-level text tokenizer (no Qwen vocabulary offline)
    speech_tokenizer/config.json, model.safetensors <- codec decoder weights under the `decoder.` prefix
"""
import os
import shutil

import numpy as np
import torch


def tiny_text_tokenizer(t):
    """Character-level fast tokenizer whose special strings map to the ids the model config names:
    <|im_start|> / <|im_end|> -> im_start/im_end ids, 'assistant' -> 77, 'user' -> 78, newline -> 198."""
    from tokenizers import Regex, Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {}
    chars = [chr(c) for c in range(32, 127)] + ["\n"]
    nxt = 0
    for ch in chars:
        if ch == "\n":
            vocab[ch] = 198
            continue
        while nxt in (77, 78, 198):
            nxt += 1
        vocab[ch] = nxt
        nxt += 1
    vocab["assistant"] = 77
    vocab["user"] = 78
    vocab["<|im_start|>"] = t.im_start_token_id
    vocab["<|im_end|>"] = t.im_end_token_id
    vocab["[UNK]"] = 199
    vocab["[PAD]"] = 200
    used = set(vocab.values())
    for i in range(t.text_vocab_size):            # a dense id space (no holes), like a real vocabulary
        if i not in used:
            vocab[f"<filler_{i}>"] = i
    tok = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Split(Regex(r"[\s\S]"), "isolated")
    tok.add_special_tokens(["<|im_start|>", "<|im_end|>"])
    tok.add_tokens(["assistant", "user"])
    return PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="[UNK]", pad_token="[PAD]")


def make_tiny_checkpoint(dst, golden_dir, t, talker_w, c, codec_w):
    from safetensors.torch import save_file
    src = os.path.join(golden_dir, "ckpt_tiny")
    os.makedirs(os.path.join(dst, "speech_tokenizer"), exist_ok=True)
    for rel in ("config.json", "generation_config.json", os.path.join("speech_tokenizer", "config.json")):
        shutil.copy(os.path.join(src, rel), os.path.join(dst, rel))
    as_t = lambda v: torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v.contiguous()
    save_file({"talker." + k: as_t(v) for k, v in talker_w.items()}, os.path.join(dst, "model.safetensors"))
    save_file({"decoder." + k: as_t(v) for k, v in codec_w.items()}, os.path.join(dst, "speech_tokenizer", "model.safetensors"))
    tiny_text_tokenizer(t).save_pretrained(dst)
    return dst
