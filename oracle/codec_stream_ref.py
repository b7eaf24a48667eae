"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Incremental (state-carrying) restatement of the Qwen3-TTS-Tokenizer-12Hz decoder: the design reference for the
streaming codec decode of SURVEY.md 8(f2).  Every layer of the decoder is causal (V2:159-208, 491), so feeding the
frames packet by packet while carrying

  * the last (k-1)*dilation input columns of every causal conv (V2:189-192),
  * the last input column of every transposed conv (k = 2*stride: output block t mixes columns t and t-1, V2:204-208),
  * the last (window-1) post-RoPE keys / values of every transformer layer (sliding window, V2:491),

reproduces `Qwen3TTSTokenizerV2Decoder.forward` (V2:869-884) on the whole sequence -- without the 25-frame re-decode
of `chunked_decode` (V2:886-896) and without its context truncation.  tests/test_oracle_golden.py checks
stream == codec_ref.decoder_forward for ragged packet sizes; `state_bytes` sizes the per-stream state for DESIGN.md.
V2 = qwen_tts/core/tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py.
"""
from typing import Dict, Optional

import torch
import torch.nn.functional as F

import codec_ref as R
from codec_ref import _t


class StreamState:
    """Per-batch streaming state: a dict of named tensors plus the number of frames consumed so far."""

    def __init__(self):
        self.t = 0                      # frames already decoded (position of the next frame)
        self.buf: Dict[str, torch.Tensor] = {}

    def state_bytes(self) -> int:
        return sum(v.numel() * v.element_size() for v in self.buf.values())


def _stream_conv(st: StreamState, key: str, x, weight, bias, dilation=1, groups=1):
    """Causal conv (stride 1) over [carried columns | new columns]; carries the last (k-1)*dilation columns."""
    k = weight.shape[-1]
    halo = (k - 1) * dilation
    prev = st.buf.get(key)
    if prev is None:
        prev = x.new_zeros(x.shape[0], x.shape[1], halo)          # == the zero left padding of V2:189-192
    xx = torch.cat([prev, x], dim=-1)
    st.buf[key] = xx[..., xx.shape[-1] - halo:] if halo > 0 else xx[..., :0]
    return F.conv1d(xx, weight, bias, dilation=dilation, groups=groups)


def _stream_transconv(st: StreamState, key: str, x, weight, bias, stride):
    """ConvTranspose1d(k, stride) followed by the reference's right trim of (k - stride).  With k == stride the op is
    column-local; with k == 2*stride output block t = W[:, :, :s] x_t + W[:, :, s:] x_{t-1}: carry one column."""
    k = weight.shape[-1]
    if k == stride:
        return F.conv_transpose1d(x, weight, bias, stride=stride)
    assert k == 2 * stride, "decoder blocks use kernel = 2 * stride (V2:647-649)"
    prev = st.buf.get(key)
    if prev is None:
        prev = x.new_zeros(x.shape[0], x.shape[1], 1)
    xx = torch.cat([prev, x], dim=-1)
    st.buf[key] = x[..., -1:].clone()
    y = F.conv_transpose1d(xx, weight, bias, stride=stride)       # ((m+1)*s + s) samples
    m = x.shape[-1]
    return y[..., stride: stride + m * stride]


def _stream_convnext(w, p, st, key, x):
    C = x.shape[1]
    h = _stream_conv(st, key + ".dw", x, _t(w, p + "dwconv.conv.weight"), _t(w, p + "dwconv.conv.bias"), groups=C)
    h = h.permute(0, 2, 1)
    h = F.layer_norm(h, (C,), _t(w, p + "norm.weight"), _t(w, p + "norm.bias"), 1e-6)
    h = F.linear(h, _t(w, p + "pwconv1.weight"), _t(w, p + "pwconv1.bias"))
    h = F.gelu(h)
    h = F.linear(h, _t(w, p + "pwconv2.weight"), _t(w, p + "pwconv2.bias"))
    h = _t(w, p + "gamma") * h
    return x + h.permute(0, 2, 1)


def _stream_res_unit(w, p, st, key, x, dilation):
    h = R.snake_beta(x, _t(w, p + "act1.alpha"), _t(w, p + "act1.beta"))
    h = _stream_conv(st, key + ".c1", h, _t(w, p + "conv1.conv.weight"), _t(w, p + "conv1.conv.bias"), dilation=dilation)
    h = R.snake_beta(h, _t(w, p + "act2.alpha"), _t(w, p + "act2.beta"))
    h = F.conv1d(h, _t(w, p + "conv2.conv.weight"), _t(w, p + "conv2.conv.bias"))       # k = 1: stateless
    return h + x


def _stream_transformer(w, cfg, st: StreamState, x):
    """V2:501-575 with a per-layer sliding KV cache: queries at global positions t0..t0+m-1 attend keys in
    (q - window, q]; the cache keeps the last window-1 positions (post-RoPE keys, values)."""
    B, m, _ = x.shape
    nh, nkv, hd, W = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.sliding_window
    p = "pre_transformer."
    t0 = st.t
    h = F.linear(x, _t(w, p + "input_proj.weight"), _t(w, p + "input_proj.bias"))
    cos, sin = R._rope_cos_sin(torch.arange(t0, t0 + m), hd, cfg.rope_theta)
    for l in range(cfg.num_hidden_layers):
        lp = f"{p}layers.{l}."
        n1 = R._rmsnorm(h, _t(w, lp + "input_layernorm.weight"), cfg.rms_norm_eps)
        q = F.linear(n1, _t(w, lp + "self_attn.q_proj.weight")).view(B, m, nh, hd).transpose(1, 2)
        k = F.linear(n1, _t(w, lp + "self_attn.k_proj.weight")).view(B, m, nkv, hd).transpose(1, 2)
        v = F.linear(n1, _t(w, lp + "self_attn.v_proj.weight")).view(B, m, nkv, hd).transpose(1, 2)
        q = q * cos + R._rotate_half(q) * sin
        k = k * cos + R._rotate_half(k) * sin
        pk, pv = st.buf.get(f"kv{l}.k"), st.buf.get(f"kv{l}.v")
        if pk is not None:
            k = torch.cat([pk, k], dim=2)
            v = torch.cat([pv, v], dim=2)
        n_prev = k.shape[2] - m                                   # cached positions: t0 - n_prev .. t0 - 1
        keep = min(W - 1, k.shape[2])
        st.buf[f"kv{l}.k"] = k[:, :, k.shape[2] - keep:].clone()
        st.buf[f"kv{l}.v"] = v[:, :, v.shape[2] - keep:].clone()
        qi = torch.arange(t0, t0 + m)[:, None]
        ki = torch.arange(t0 - n_prev, t0 + m)[None, :]
        allowed = (ki <= qi) & (ki > qi - W)
        bias = torch.zeros(m, n_prev + m).masked_fill(~allowed, float("-inf"))
        kk, vv = k, v
        if nkv != nh:
            kk = kk.repeat_interleave(nh // nkv, dim=1)
            vv = vv.repeat_interleave(nh // nkv, dim=1)
        a = torch.matmul(q, kk.transpose(2, 3)) * (hd ** -0.5) + bias
        a = torch.softmax(a, dim=-1, dtype=torch.float32)
        o = torch.matmul(a, vv).transpose(1, 2).reshape(B, m, nh * hd)
        o = F.linear(o, _t(w, lp + "self_attn.o_proj.weight"))
        h = h + _t(w, lp + "self_attn_layer_scale.scale") * o
        n2 = R._rmsnorm(h, _t(w, lp + "post_attention_layernorm.weight"), cfg.rms_norm_eps)
        mm = F.linear(F.silu(F.linear(n2, _t(w, lp + "mlp.gate_proj.weight"))) *
                      F.linear(n2, _t(w, lp + "mlp.up_proj.weight")), _t(w, lp + "mlp.down_proj.weight"))
        h = h + _t(w, lp + "mlp_layer_scale.scale") * mm
    h = R._rmsnorm(h, _t(w, p + "norm.weight"), cfg.rms_norm_eps)
    return F.linear(h, _t(w, p + "output_proj.weight"), _t(w, p + "output_proj.bias"))


def decoder_step(w, cfg, codes: torch.Tensor, st: Optional[StreamState] = None):
    """Decode the next `m` frames of a stream: codes (B, Q, m) int64 -> (waveform (B, 1, m*total_upsample), state).
    Concatenating the outputs of successive calls equals codec_ref.decoder_forward on the concatenated codes."""
    if codes.shape[1] != cfg.num_quantizers:
        raise ValueError(f"Expected {cfg.num_quantizers} layer of codes, got {codes.shape[1]}")
    st = st if st is not None else StreamState()
    h = R.rvq_dequant(w, cfg, codes)
    h = _stream_conv(st, "pre_conv", h, _t(w, "pre_conv.conv.weight"), _t(w, "pre_conv.conv.bias"))
    h = _stream_transformer(w, cfg, st, h.transpose(1, 2)).permute(0, 2, 1)
    for u, f in enumerate(cfg.upsampling_ratios):
        h = _stream_transconv(st, f"up{u}.t", h, _t(w, f"upsample.{u}.0.conv.weight"), _t(w, f"upsample.{u}.0.conv.bias"), f)
        h = _stream_convnext(w, f"upsample.{u}.1.", st, f"up{u}.cn", h)
    h = _stream_conv(st, "dec0", h, _t(w, "decoder.0.conv.weight"), _t(w, "decoder.0.conv.bias"))
    for i, r in enumerate(cfg.upsample_rates):
        p = f"decoder.{i + 1}.block."
        h = R.snake_beta(h, _t(w, p + "0.alpha"), _t(w, p + "0.beta"))
        h = _stream_transconv(st, f"blk{i}.t", h, _t(w, p + "1.conv.weight"), _t(w, p + "1.conv.bias"), r)
        for j, d in zip((2, 3, 4), (1, 3, 9)):
            h = _stream_res_unit(w, p + f"{j}.", st, f"blk{i}.ru{j}", h, d)
    n = len(cfg.upsample_rates)
    h = R.snake_beta(h, _t(w, f"decoder.{n + 1}.alpha"), _t(w, f"decoder.{n + 1}.beta"))
    h = _stream_conv(st, "final", h, _t(w, f"decoder.{n + 2}.conv.weight"), _t(w, f"decoder.{n + 2}.conv.bias"))
    st.t += codes.shape[-1]
    return h.clamp(min=-1, max=1), st
