"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU fp32 restatement of the Qwen3-TTS autoregressive speech-token decoder: the talker
(prefill + per-frame decode), the nested 15-pass code predictor, the HF logits
processors / sampler, the outer generate loop and the prompt assembly.

Functional torch code over a flat {name: tensor} weight dict whose names are the
reference state_dict names relative to `talker.`.
M = qwen_tts/core/models/modeling_qwen3_tts.py.  The outer loop is third-party
(`transformers==4.57.3` GenerationMixin._sample, pyproject.toml:23) and is restated here
from its published behaviour.  Pinned against the reference's own modules
by tests/golden/talker_*.npz (oracle/gen_golden.py).
"""
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn.functional as F


def _t(w, k):
    v = w[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


# ----------------------------------------------------------------------------- primitives
def rmsnorm(x, weight, eps):
    """Qwen3TTSRMSNorm.forward M:605-610 (fp32 inside, cast to the input dtype BEFORE x weight)."""
    dt = x.dtype
    xf = x.to(torch.float32)
    v = xf.pow(2).mean(-1, keepdim=True)
    return weight * (xf * torch.rsqrt(v + eps)).to(dt)


def rope_cos_sin(positions, head_dim, theta, dtype=torch.float32):
    """Qwen3TTS(Talker)RotaryEmbedding.forward M:546-559 / M:581-592 ('default' rope: inv_freq =
    theta^(-2i/d), scaling 1).  The talker's 3-row M-RoPE collapses to this because get_rope_index
    (M:1794-1796) emits three identical rows.  positions: (...,) float/int."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    fr = positions.float()[..., None] * inv
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def attention(q, k, v, bias):
    """eager_attention_forward M:634-657.  q (B,nh,Tq,hd), k/v (B,nkv,S,hd), bias (B,1,Tq,S) additive."""
    nh, nkv = q.shape[1], k.shape[1]
    if nkv != nh:
        k = k.repeat_interleave(nh // nkv, dim=1)
        v = v.repeat_interleave(nh // nkv, dim=1)
    a = torch.matmul(q, k.transpose(2, 3)) * (q.shape[-1] ** -0.5) + bias
    a = torch.softmax(a, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(a, v).transpose(1, 2)


class KV:
    """Grow-by-append KV store with DynamicCache.update semantics (append along seq)."""

    def __init__(self, n_layers):
        self.k = [None] * n_layers
        self.v = [None] * n_layers

    def update(self, l, k, v):
        self.k[l] = k if self.k[l] is None else torch.cat((self.k[l], k), dim=2)
        self.v[l] = v if self.v[l] is None else torch.cat((self.v[l], v), dim=2)
        return self.k[l], self.v[l]

    def length(self):
        return 0 if self.k[0] is None else self.k[0].shape[2]


def decoder_stack(w, prefix, n_layers, nh, nkv, hd, eps, x, cos, sin, bias, kv: KV):
    """N x pre-norm residual block (M:1393-1424 talker / M:985-1012 code predictor) with q/k RMSNorm
    over head_dim applied before RoPE (M:773-780 / M:928-933).  x (B,T,H)."""
    B, T, _ = x.shape
    for l in range(n_layers):
        p = f"{prefix}layers.{l}."
        n1 = rmsnorm(x, _t(w, p + "input_layernorm.weight"), eps)
        q = F.linear(n1, _t(w, p + "self_attn.q_proj.weight")).view(B, T, nh, hd)
        k = F.linear(n1, _t(w, p + "self_attn.k_proj.weight")).view(B, T, nkv, hd)
        v = F.linear(n1, _t(w, p + "self_attn.v_proj.weight")).view(B, T, nkv, hd)
        q = rmsnorm(q, _t(w, p + "self_attn.q_norm.weight"), eps).transpose(1, 2)
        k = rmsnorm(k, _t(w, p + "self_attn.k_norm.weight"), eps).transpose(1, 2)
        v = v.transpose(1, 2)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        kk, vv = kv.update(l, k, v)
        o = attention(q, kk, vv, bias).reshape(B, T, nh * hd)
        x = x + F.linear(o, _t(w, p + "self_attn.o_proj.weight"))
        n2 = rmsnorm(x, _t(w, p + "post_attention_layernorm.weight"), eps)
        m = F.linear(F.silu(F.linear(n2, _t(w, p + "mlp.gate_proj.weight"))) *
                     F.linear(n2, _t(w, p + "mlp.up_proj.weight")), _t(w, p + "mlp.down_proj.weight"))
        x = x + m
    return x


# ----------------------------------------------------------------------------- sampling (HF semantics)
@dataclass
class SamplingParams:
    do_sample: bool = True
    top_k: Optional[int] = 50
    top_p: Optional[float] = 1.0
    temperature: Optional[float] = 0.9
    repetition_penalty: float = 1.05
    subtalker_dosample: bool = True
    subtalker_top_k: Optional[int] = 50
    subtalker_top_p: Optional[float] = 1.0
    subtalker_temperature: Optional[float] = 0.9


def process_logits(scores, generated, *, repetition_penalty=1.0, eos_id=None, min_new_tokens=0,
                   suppress=None, do_sample=False, temperature=None, top_k=None, top_p=None):
    """HF processor chain in `_get_logits_processor` order (transformers 4.57.3, generation/utils.py):
    RepetitionPenalty -> MinNewTokensLength -> SuppressTokens -> [Temperature -> TopK -> TopP].
    scores (B,V) fp32, generated (B,t) int64 = tokens sampled so far (prompt is embeds-only so the
    penalty sees generated ids only)."""
    scores = scores.clone()
    if repetition_penalty is not None and repetition_penalty != 1.0 and generated.shape[1] > 0:
        s = torch.gather(scores, 1, generated)
        s = torch.where(s < 0, s * repetition_penalty, s / repetition_penalty)
        scores = scores.scatter(1, generated, s)
    if eos_id is not None and min_new_tokens > 0 and generated.shape[1] < min_new_tokens:
        scores[:, eos_id] = float("-inf")
    if suppress is not None and len(suppress) > 0:
        scores[:, suppress] = float("-inf")
    if do_sample:
        if temperature is not None and temperature != 1.0:
            scores = scores / temperature
        if top_k is not None and top_k != 0:
            kk = min(top_k, scores.shape[-1])
            thr = torch.topk(scores, kk)[0][..., -1, None]
            scores = scores.masked_fill(scores < thr, float("-inf"))
        if top_p is not None and top_p < 1.0:
            ss, si = torch.sort(scores, descending=False)
            cp = ss.softmax(dim=-1).cumsum(dim=-1)
            rm = cp <= (1 - top_p)
            rm[..., -1:] = False
            scores = scores.masked_fill(rm.scatter(1, si, rm), float("-inf"))
    return scores


def pick(scores, do_sample, generator=None):
    """argmax (first max wins, like torch.argmax) or softmax + multinomial."""
    if do_sample:
        return torch.multinomial(F.softmax(scores, dim=-1), 1, generator=generator).squeeze(1)
    return torch.argmax(scores, dim=-1)


# ----------------------------------------------------------------------------- code predictor
def code_predictor_generate(w, cfg, past_hidden, last_id_hidden, sp: SamplingParams, generator=None,
                            trace: dict = None):
    """talker.code_predictor.generate(inputs_embeds=cat(past_hidden,last_id_hidden), max_new_tokens=G-1,
    do_sample=subtalker_*) M:1671-1680 -> Qwen3TTSTalkerCodePredictorModelForConditionalGeneration.forward
    M:1250-1312: pass 0 on 2 tokens with lm_head[0]; pass j on codec_embedding[j-1](token) with lm_head[j].
    Returns (B, G-1) int64."""
    G = cfg.num_code_groups
    B = past_hidden.shape[0]
    nh, nkv, hd = cfg.cp_num_attention_heads, cfg.cp_num_key_value_heads, cfg.cp_head_dim
    pre = "code_predictor.model."
    kv = KV(cfg.cp_num_hidden_layers)
    x = torch.cat((past_hidden, last_id_hidden), dim=1)                 # (B,2,H)
    toks = []
    for j in range(G - 1):
        if j > 0:
            x = F.embedding(toks[-1][:, None], _t(w, f"{pre}codec_embedding.{j - 1}.weight"))   # M:1281
        if "code_predictor.small_to_mtp_projection.weight" in w:
            x = F.linear(x, _t(w, "code_predictor.small_to_mtp_projection.weight"),
                         _t(w, "code_predictor.small_to_mtp_projection.bias"))                   # M:1282
        T = x.shape[1]
        s0 = kv.length()
        pos = torch.arange(s0, s0 + T)
        cos, sin = rope_cos_sin(pos, hd, cfg.cp_rope_theta)
        qi = pos[:, None]
        ki = torch.arange(s0 + T)[None, :]
        bias = torch.zeros(T, s0 + T).masked_fill(ki > qi, float("-inf"))[None, None]
        h = decoder_stack(w, pre, cfg.cp_num_hidden_layers, nh, nkv, hd, cfg.cp_rms_norm_eps, x,
                          cos, sin, bias, kv)
        h = rmsnorm(h, _t(w, pre + "norm.weight"), cfg.cp_rms_norm_eps)
        logits = F.linear(h[:, -1], _t(w, f"code_predictor.lm_head.{j}.weight")).float()          # M:1299
        sc = process_logits(logits, torch.zeros(B, 0, dtype=torch.long), do_sample=sp.subtalker_dosample,
                            temperature=sp.subtalker_temperature, top_k=sp.subtalker_top_k,
                            top_p=sp.subtalker_top_p)
        if trace is not None:
            trace.setdefault("cp_logits", []).append(logits.clone())
        toks.append(pick(sc, sp.subtalker_dosample, generator))
    return torch.stack(toks, dim=1)


# ----------------------------------------------------------------------------- talker generate (seam S2)
def talker_generate(w, cfg, inputs_embeds, attention_mask, trailing_text_hidden, tts_pad_embed,
                    max_new_tokens=2048, min_new_tokens=2, eos_token_id=None, suppress_tokens=None,
                    sp: SamplingParams = None, generator=None, trace: dict = None):
    """`self.talker.generate(inputs_embeds, attention_mask, trailing_text_hidden, tts_pad_embed,
    **talker_kwargs)` M:2272-2278 = HF `_sample` around Qwen3TTSTalkerForConditionalGeneration.forward
    M:1636-1744.

    inputs_embeds (B,T,H) LEFT-padded with zeros, attention_mask (B,T) {0,1}, trailing_text_hidden
    (B,Tt,H), tts_pad_embed (1,1,H).
    Returns dict(codes (B, n_frames, G) int64 [every forwarded frame, untrimmed],
                 tokens (B, n_tokens) int64, hidden (B, n_frames, H) [past_hidden per frame]).
    """
    sp = sp or SamplingParams()
    eos = cfg.codec_eos_token_id if eos_token_id is None else eos_token_id
    if suppress_tokens is None:
        suppress_tokens = [i for i in range(cfg.vocab_size - 1024, cfg.vocab_size) if i != cfg.codec_eos_token_id]  # M:2059-2063
    B, T, H = inputs_embeds.shape
    nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    G = cfg.num_code_groups
    pre = "model."
    kv = KV(cfg.num_hidden_layers)
    mask = attention_mask.to(torch.long)
    n_pad = (1 - mask).sum(-1)                                           # rope_deltas = -n_pad (M:1699-1704)

    # ---- prefill (M:1665-1667, positions M:1794-1796: cumsum(mask)-1, pads -> 1)
    pos = (mask.float().cumsum(-1) - 1).masked_fill(mask == 0, 1)
    cos, sin = rope_cos_sin(pos, hd, cfg.rope_theta)
    cos, sin = cos[:, None], sin[:, None]
    qi = torch.arange(T)[None, :, None]
    ki = torch.arange(T)[None, None, :]
    allowed = (ki <= qi) & (mask[:, None, :] == 1)
    bias = torch.zeros(B, T, T).masked_fill(~allowed, torch.finfo(torch.float32).min)[:, None]
    h = decoder_stack(w, pre, cfg.num_hidden_layers, nh, nkv, hd, cfg.rms_norm_eps, inputs_embeds,
                      cos, sin, bias, kv)
    h = rmsnorm(h, _t(w, pre + "norm.weight"), cfg.rms_norm_eps)
    past_hidden = h[:, -1:, :]                                           # M:1740
    logits = F.linear(past_hidden[:, 0], _t(w, "codec_head.weight")).float()        # M:1727
    generation_step = 0                                                  # M:1666 (-1) + 1 (M:1741)

    generated = torch.zeros(B, 0, dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)
    frames, hiddens = [], []
    key_mask = mask.clone()
    while True:
        sc = process_logits(logits, generated, repetition_penalty=sp.repetition_penalty, eos_id=eos,
                            min_new_tokens=min_new_tokens, suppress=suppress_tokens, do_sample=sp.do_sample,
                            temperature=sp.temperature, top_k=sp.top_k, top_p=sp.top_p)
        if trace is not None:
            trace.setdefault("logits", []).append(logits.clone())
            trace.setdefault("scores", []).append(sc.clone())
        tok = pick(sc, sp.do_sample, generator)
        tok = tok * unfinished + eos * (1 - unfinished)                  # pad_token_id falls back to eos
        generated = torch.cat((generated, tok[:, None]), dim=1)
        unfinished = unfinished & (tok != eos).long()
        if generated.shape[1] >= max_new_tokens or unfinished.max() == 0:
            break
        # ---- decode forward (M:1669-1692)
        last_id_hidden = F.embedding(tok[:, None], _t(w, pre + "codec_embedding.weight"))
        sub = code_predictor_generate(w, cfg, past_hidden, last_id_hidden, sp, generator, trace)
        codec_ids = torch.cat((tok[:, None], sub), dim=-1)               # M:1681
        emb = last_id_hidden[:, 0]
        for i in range(G - 1):
            emb = emb + F.embedding(sub[:, i], _t(w, f"code_predictor.model.codec_embedding.{i}.weight"))
        # NOTE reference sums via cat(...).sum(1) (M:1682-1687): same terms, pairwise-vs-sequential
        # fp32 summation order may differ in the last ulp; the golden comparison tolerates that.
        if generation_step < trailing_text_hidden.shape[1]:
            emb = emb + trailing_text_hidden[:, generation_step]
        else:
            emb = emb + tts_pad_embed[0, 0]
        frames.append(codec_ids)
        hiddens.append(past_hidden[:, 0])
        S = kv.length()
        key_mask = torch.cat((key_mask, torch.ones(B, 1, dtype=torch.long)), dim=1)
        pos = (S - n_pad).float()[:, None]                               # M:1706-1710
        cos, sin = rope_cos_sin(pos, hd, cfg.rope_theta)
        cos, sin = cos[:, None], sin[:, None]
        bias = torch.zeros(B, 1, S + 1).masked_fill(key_mask[:, None, :] == 0, torch.finfo(torch.float32).min)[:, None]
        h = decoder_stack(w, pre, cfg.num_hidden_layers, nh, nkv, hd, cfg.rms_norm_eps, emb[:, None],
                          cos, sin, bias, kv)
        h = rmsnorm(h, _t(w, pre + "norm.weight"), cfg.rms_norm_eps)
        past_hidden = h[:, -1:, :]
        logits = F.linear(past_hidden[:, 0], _t(w, "codec_head.weight")).float()
        generation_step += 1
    codes = torch.stack(frames, dim=1) if frames else torch.zeros(B, 0, G, dtype=torch.long)
    hidden = torch.stack(hiddens, dim=1) if hiddens else torch.zeros(B, 0, H)
    return {"codes": codes, "tokens": generated, "hidden": hidden}


def trim_at_eos(codes, eos):
    """M:2283-2289: cut each row at the first frame whose codebook-0 id is eos."""
    out = []
    for row in codes:
        is_stop = row[:, 0] == eos
        n = int(torch.argmax(is_stop.int())) if bool(is_stop.any()) else row.shape[0]
        out.append(row[:n])
    return out


# ----------------------------------------------------------------------------- prompt assembly (seam S1)
def text_projection(w, x):
    """Qwen3TTSTalkerResizeMLP M:808-816 (bias=True, act = hidden_act 'silu', M:1575-1577)."""
    h = F.linear(x, _t(w, "text_projection.linear_fc1.weight"), _t(w, "text_projection.linear_fc1.bias"))
    return F.linear(F.silu(h), _t(w, "text_projection.linear_fc2.weight"), _t(w, "text_projection.linear_fc2.bias"))


def assemble_prompts(w, cfg, input_ids: List[torch.Tensor], languages: List[str],
                     speakers: Optional[List[Optional[str]]] = None,
                     instruct_ids: Optional[List[Optional[torch.Tensor]]] = None,
                     non_streaming_mode: bool = False,
                     ref_ids=None, voice_clone_prompt: dict = None):
    """Qwen3TTSForConditionalGeneration.generate M:2068-2269: per-request prefill embeddings, LEFT pad,
    attention mask, right-padded trailing text.  Returns (embeds (B,T,H), mask (B,T), trailing (B,Tt,H),
    tts_pad_embed (1,1,H))."""
    text_emb = lambda ids: F.embedding(ids, _t(w, "model.text_embedding.weight"))
    codec_emb = lambda ids: F.embedding(ids, _t(w, "model.codec_embedding.weight"))
    ids = lambda lst: torch.tensor(lst, dtype=torch.long)
    G = cfg.num_code_groups
    per_req = [[] for _ in input_ids]
    spk_embeds = None
    if voice_clone_prompt is not None:                                   # M:1957-1966
        spk_embeds = [e.float() for e in voice_clone_prompt["ref_spk_embedding"]]
    if instruct_ids is not None:                                         # M:2076-2080
        for i, ins in enumerate(instruct_ids):
            if ins is not None:
                per_req[i].append(text_projection(w, text_emb(ins)))
    if speakers is None:
        speakers = [None] * len(input_ids)
    trailing = []
    tts_pad_embed = None
    for i, (iid, language, speaker) in enumerate(zip(input_ids, languages, speakers)):
        if spk_embeds is None:
            if speaker == "" or speaker is None:
                speaker_embed = None
            else:
                if speaker.lower() not in cfg.spk_id:
                    raise NotImplementedError(f"Speaker {speaker} not implemented")           # M:2092
                speaker_embed = codec_emb(ids(cfg.spk_id[speaker.lower()]))
        else:
            if voice_clone_prompt["x_vector_only_mode"][i] or voice_clone_prompt["icl_mode"][i]:
                speaker_embed = spk_embeds[i]
            else:
                speaker_embed = None
        assert language is not None
        if language.lower() == "auto":
            language_id = None
        else:
            if language.lower() not in cfg.codec_language_id:
                raise NotImplementedError(f"Language {language} not implemented")             # M:2114
            language_id = cfg.codec_language_id[language.lower()]
        if (language.lower() in ["chinese", "auto"] and speaker != "" and speaker is not None
                and cfg.spk_is_dialect[speaker.lower()] is not False):                       # M:2118-2122
            language_id = cfg.codec_language_id[cfg.spk_is_dialect[speaker.lower()]]
        bos_e, eos_e, pad_e = text_projection(w, text_emb(ids([[cfg.tts_bos_token_id, cfg.tts_eos_token_id,
                                                                cfg.tts_pad_token_id]]))).chunk(3, dim=1)
        tts_pad_embed = pad_e
        if language_id is None:                                          # M:2135-2147
            pre = [[cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id]]
        else:
            pre = [[cfg.codec_think_id, cfg.codec_think_bos_id, language_id, cfg.codec_think_eos_id]]
        c0 = codec_emb(ids(pre))
        c1 = codec_emb(ids([[cfg.codec_pad_id, cfg.codec_bos_id]]))
        cin = torch.cat([c0, c1], dim=1) if speaker_embed is None else \
            torch.cat([c0, speaker_embed.view(1, 1, -1), c1], dim=1)     # M:2166-2172
        role = text_projection(w, text_emb(iid[:, :3]))                  # M:2177
        body = torch.cat((pad_e.expand(-1, cin.shape[1] - 2, -1), bos_e), dim=1) + cin[:, :-1]   # M:2182-2184
        emb = torch.cat((role, body), dim=1)
        if (voice_clone_prompt is not None and voice_clone_prompt.get("ref_code") is not None
                and voice_clone_prompt["icl_mode"][i]):
            icl, trail = icl_prompt(w, cfg, iid[:, 3:-5], ref_ids[i][:, 3:-2], voice_clone_prompt["ref_code"][i],
                                    pad_e, eos_e, non_streaming_mode)
            emb = torch.cat([emb, icl], dim=1)
        else:
            emb = torch.cat([emb, text_projection(w, text_emb(iid[:, 3:4])) + cin[:, -1:]], dim=1)   # M:2200-2202
            if non_streaming_mode:                                       # M:2203-2227
                emb = emb[:, :-1]
                n_txt = iid[:, 3:-5].shape[1]
                emb = torch.cat([emb,
                                 torch.cat((text_projection(w, text_emb(iid[:, 3:-5])), eos_e), dim=1)
                                 + codec_emb(ids([[cfg.codec_pad_id] * (n_txt + 1)])),
                                 pad_e + codec_emb(ids([[cfg.codec_bos_id]]))], dim=1)
                trail = pad_e
            else:                                                        # M:2229-2232
                trail = torch.cat((text_projection(w, text_emb(iid[:, 4:-5])), eos_e), dim=1)
        per_req[i].append(emb)
        trailing.append(trail)
    seqs = [torch.cat(p, dim=1).squeeze(0) for p in per_req]
    lens = torch.tensor([s.shape[0] for s in seqs])
    Tm = int(lens.max())
    H = seqs[0].shape[-1]
    embeds = torch.zeros(len(seqs), Tm, H)
    for i, s in enumerate(seqs):                                         # LEFT pad with zeros M:2239-2249
        embeds[i, Tm - s.shape[0]:] = s
    mask = (torch.arange(Tm)[None, :] >= (Tm - lens)[:, None]).long()    # M:2251-2254
    tl = [t.squeeze(0) for t in trailing]
    Tt = max(t.shape[0] for t in tl)
    tr = pad_e.squeeze()[None, None, :].repeat(len(tl), Tt, 1).clone()   # right pad with tts_pad M:2255-2269
    for i, t in enumerate(tl):
        tr[i, : t.shape[0]] = t
    return embeds, mask, tr, tts_pad_embed


def icl_prompt(w, cfg, text_id, ref_id, ref_code, pad_e, eos_e, non_streaming_mode):
    """generate_icl_prompt M:1968-2019."""
    text_emb = lambda ids: F.embedding(ids, _t(w, "model.text_embedding.weight"))
    codec_emb = lambda ids: F.embedding(ids, _t(w, "model.codec_embedding.weight"))
    te = torch.cat([text_projection(w, text_emb(torch.cat([ref_id, text_id], dim=-1))), eos_e], dim=1)
    parts = []
    for i in range(cfg.num_code_groups):
        if i == 0:
            parts.append(codec_emb(ref_code[:, :1]))
        else:
            parts.append(F.embedding(ref_code[:, i:i + 1], _t(w, f"code_predictor.model.codec_embedding.{i - 1}.weight")))
    ce = torch.cat(parts, dim=1).sum(1).unsqueeze(0)
    ce = torch.cat([codec_emb(torch.tensor([[cfg.codec_bos_id]])), ce], dim=1)
    tl, cl = te.shape[1], ce.shape[1]
    if non_streaming_mode:
        icl = te + codec_emb(torch.tensor([[cfg.codec_pad_id] * tl]))
        return torch.cat([icl, ce + pad_e], dim=1), pad_e
    if tl > cl:
        return te[:, :cl] + ce, te[:, cl:]
    te = torch.cat([te] + [pad_e] * (cl - tl), dim=1)
    return te + ce, pad_e


def generate(w, cfg, input_ids, languages, speakers=None, instruct_ids=None, non_streaming_mode=False,
             max_new_tokens=4096, sp: SamplingParams = None, generator=None, ref_ids=None,
             voice_clone_prompt=None, eos_token_id=None):
    """Qwen3TTSForConditionalGeneration.generate M:2022-2292 -> (list[(Ti,G) int64], list[(Ti,H)])."""
    emb, mask, tr, pad = assemble_prompts(w, cfg, input_ids, languages, speakers, instruct_ids,
                                          non_streaming_mode, ref_ids, voice_clone_prompt)
    r = talker_generate(w, cfg, emb, mask, tr, pad, max_new_tokens=max_new_tokens, min_new_tokens=2,
                        eos_token_id=eos_token_id, sp=sp, generator=generator)
    codes = trim_at_eos(r["codes"], cfg.codec_eos_token_id)
    return codes, [h[: c.shape[0]] for h, c in zip(r["hidden"], codes)]
