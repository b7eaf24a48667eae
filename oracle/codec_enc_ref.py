"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU fp32 restatement of the Qwen3-TTS-Tokenizer-12Hz *encoder* (waveform -> codes), SURVEY.md 8(f3): the oracle for a
later HIP implementation of `Qwen3TTSTokenizer.encode` / `create_voice_clone_prompt`.

The reference's encoder IS a third-party model: `Qwen3TTSTokenizerV2Encoder(MimiModel)` (tokenizer v2:897-908) and
`Qwen3TTSTokenizerV2Model.encode` (v2:961-991) call `transformers.MimiModel.encode` (reference pins transformers
4.57.3, pyproject.toml:23; the build container has 5.15.0 -- same Mimi classes).  The algorithm restated here is the
published one (Moshi/Mimi: SEANet encoder -> 8-layer causal transformer -> stride-2 downsample -> split residual VQ),
following transformers/models/mimi/modeling_mimi.py (TM below); parity is anchored on the reference's call site
(v2:982-985: codes[:, :encoder_valid_num_quantizers], trimmed to ceil(n_samples / encode_downsample_rate) frames) and
pinned against the reference's own encoder class run here (tests/golden/codec_enc_tiny.npz, oracle/gen_golden.py).

Weights: flat {name: tensor} dict, reference state_dict names relative to `encoder.`.
"""
import math
from typing import List

import torch
import torch.nn.functional as F


def _t(w, k):
    v = w[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def mimi_conv1d(x, weight, bias, stride=1, dilation=1, pad_mode="constant"):
    """MimiConv1d.forward, causal branch (TM:210-347): left pad `padding_total` = effective kernel - stride, right pad
    whatever makes the last window full (`_get_extra_padding_for_conv1d`), then a plain strided conv."""
    k_eff = (weight.shape[-1] - 1) * dilation + 1
    p_total = k_eff - stride
    L = x.shape[-1]
    n_frames = math.ceil((L - k_eff + p_total) / stride + 1) - 1
    extra = n_frames * stride + k_eff - p_total - L
    x = F.pad(x, (p_total, extra), mode=pad_mode) if pad_mode != "constant" else F.pad(x, (p_total, extra))
    return F.conv1d(x, weight, bias, stride=stride, dilation=dilation)


def seanet_encoder(w, cfg, x):
    """MimiEncoder (TM:450-492): conv k7 -> for each ratio (reversed): residual block(s) -> ELU -> strided conv (k = 2r,
    channels x2) -> ELU -> conv k3 to hidden_size.  Residual block (TM:408-447): ELU, conv k3 (dilation g^j) to
    dim/compress, ELU, conv k1 back to dim, identity shortcut."""
    p = "encoder.layers."
    idx = 0
    h = mimi_conv1d(x, _t(w, f"{p}{idx}.conv.weight"), _t(w, f"{p}{idx}.conv.bias"))
    idx += 1
    for ratio in reversed(cfg.upsampling_ratios):
        for j in range(cfg.num_residual_layers):
            r = h
            h = F.elu(h)
            h = mimi_conv1d(h, _t(w, f"{p}{idx}.block.1.conv.weight"), _t(w, f"{p}{idx}.block.1.conv.bias"),
                            dilation=cfg.dilation_growth_rate ** j)
            h = F.elu(h)
            h = mimi_conv1d(h, _t(w, f"{p}{idx}.block.3.conv.weight"), _t(w, f"{p}{idx}.block.3.conv.bias"))
            h = r + h
            idx += 1
        h = F.elu(h)
        idx += 1                                       # the ELU is a module of the ModuleList too
        h = mimi_conv1d(h, _t(w, f"{p}{idx}.conv.weight"), _t(w, f"{p}{idx}.conv.bias"), stride=ratio)
        idx += 1
    h = F.elu(h)
    idx += 1
    return mimi_conv1d(h, _t(w, f"{p}{idx}.conv.weight"), _t(w, f"{p}{idx}.conv.bias"))


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def encoder_transformer(w, cfg, x):
    """MimiTransformerModel (TM:782-928) without cache: pre-LayerNorm blocks, MHA with RoPE under a sliding-window
    causal mask, LayerScale on both residual branches, GELU MLP without biases."""
    B, T, _ = x.shape
    nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    fr = torch.arange(T).float()[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), dim=-1)
    cos, sin = emb.cos(), emb.sin()
    qi, ki = torch.arange(T)[:, None], torch.arange(T)[None, :]
    bias = torch.zeros(T, T).masked_fill(~((ki <= qi) & (ki > qi - cfg.sliding_window)), float("-inf"))
    h = x
    for l in range(cfg.num_hidden_layers):
        lp = f"encoder_transformer.layers.{l}."
        n1 = F.layer_norm(h, (h.shape[-1],), _t(w, lp + "input_layernorm.weight"), _t(w, lp + "input_layernorm.bias"), cfg.norm_eps)
        q = F.linear(n1, _t(w, lp + "self_attn.q_proj.weight")).view(B, T, nh, hd).transpose(1, 2)
        k = F.linear(n1, _t(w, lp + "self_attn.k_proj.weight")).view(B, T, nkv, hd).transpose(1, 2)
        v = F.linear(n1, _t(w, lp + "self_attn.v_proj.weight")).view(B, T, nkv, hd).transpose(1, 2)
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        if nkv != nh:
            k = k.repeat_interleave(nh // nkv, dim=1)
            v = v.repeat_interleave(nh // nkv, dim=1)
        a = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5) + bias
        a = torch.softmax(a, dim=-1, dtype=torch.float32)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, nh * hd)
        o = F.linear(o, _t(w, lp + "self_attn.o_proj.weight"))
        h = h + _t(w, lp + "self_attn_layer_scale.scale") * o
        n2 = F.layer_norm(h, (h.shape[-1],), _t(w, lp + "post_attention_layernorm.weight"),
                          _t(w, lp + "post_attention_layernorm.bias"), cfg.norm_eps)
        m = F.linear(F.gelu(F.linear(n2, _t(w, lp + "mlp.fc1.weight"))), _t(w, lp + "mlp.fc2.weight"))
        h = h + _t(w, lp + "mlp_layer_scale.scale") * m
    return h


def rvq_encode(w, prefix, x, n_layers, margins=None):
    """MimiResidualVectorQuantizer.encode (TM:1050-1068): 1x1 input_proj, then per layer nearest codebook entry
    (Euclidean, argmin of cdist; embed = embed_sum / clamp(cluster_usage, 1e-5), TM:964-1007) and residual update."""
    r = F.conv1d(x, _t(w, prefix + "input_proj.weight"))
    out = []
    for i in range(n_layers):
        es = _t(w, f"{prefix}layers.{i}.codebook.embed_sum")
        cu = _t(w, f"{prefix}layers.{i}.codebook.cluster_usage")
        table = es / cu.clamp(min=1e-5)[:, None]
        flat = r.permute(0, 2, 1).reshape(-1, r.shape[1])
        dist = torch.cdist(flat[None].float(), table[None].float(), p=2)[0]
        ind = dist.argmin(dim=-1).view(r.shape[0], r.shape[2])
        if margins is not None:          # (test infrastructure: how far the runner-up entry is, relative to the winner -- the near-tie exemption of the parity tests)
            d2 = torch.topk(dist, 2, dim=-1, largest=False).values
            margins.append(((d2[:, 1] - d2[:, 0]) / d2[:, 0].clamp(min=1e-12)).view(r.shape[0], r.shape[2]))
        r = r - F.embedding(ind, table).permute(0, 2, 1)
        out.append(ind)
    return torch.stack(out)                                        # (layers, B, T)


def mimi_encode(w, cfg, wav: torch.Tensor, num_quantizers: int = None, margins=None) -> torch.Tensor:
    """MimiModel.encode / _encode_frame (TM:1230-1262, 1297-1394), non-streaming.  wav (B, 1, samples) ->
    codes (B, num_quantizers, frames) int64."""
    nq = cfg.num_quantizers if num_quantizers is None else num_quantizers
    h = seanet_encoder(w, cfg, wav)
    h = encoder_transformer(w, cfg, h.transpose(1, 2)).transpose(1, 2)
    h = mimi_conv1d(h, _t(w, "downsample.conv.weight"), None, stride=2, pad_mode="replicate")
    ns = cfg.num_semantic_quantizers
    codes = rvq_encode(w, "quantizer.semantic_residual_vector_quantizer.", h, ns, margins)
    if nq > ns:
        codes = torch.cat([codes, rvq_encode(w, "quantizer.acoustic_residual_vector_quantizer.", h, nq - ns, margins)], dim=0)
    return codes.transpose(0, 1)                                   # (margins: a list of nq tensors (B, T), in codebook order)


def model_encode(w, cfg, input_values: torch.Tensor, padding_mask: torch.Tensor) -> List[torch.Tensor]:
    """Qwen3TTSTokenizerV2Model.encode (tokenizer v2:961-991): input_values (B, samples) zero-padded, padding_mask
    (B, samples) {0,1}.  Returns per-row codes (frames_i, encoder_valid_num_quantizers) with
    frames_i = ceil(valid_samples_i / encode_downsample_rate)."""
    codes = mimi_encode(w, cfg, input_values.unsqueeze(1))[:, :cfg.encoder_valid_num_quantizers]
    out = []
    for c, m in zip(codes, padding_mask):
        n = -(-int(m.sum()) // cfg.encode_downsample_rate)
        out.append(c[..., :n].transpose(0, 1))
    return out
