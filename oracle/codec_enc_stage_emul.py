"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Python mirror of `qtts_encoder::encode` (qwen3-tts_amd/csrc/encoder_engine.hip): the same channel-last staging --
stride-1 causal convs as zero-left-padded tap GEMMs, strided convs (k = 2r, stride r) as 2-tap GEMMs over "super-rows"
of r consecutive rows after zero-padding the length to a multiple of r, the replicate-padded downsample with one
leading super-row that is dropped afterwards, and the codebook search as argmin(||e||^2 - 2 r.e) on GEMM scores --
checked against oracle/codec_enc_ref.py (which is pinned to the reference's encoder class) in tests/test_oracle_golden.py.
It verifies the ALGEBRA of the C++ orchestration; the GEMM / attention kernels it stands in for are the validated ones."""
import torch
import torch.nn.functional as F

import codec_enc_ref as E
from codec_enc_ref import _t

def tap_gemm(x, T, Wt, bias, shifts):
    """Model of gemm_tap: x (B*T, K) channel-last rows, Wt (taps, N, K); out[m] = sum_tap W[tap] @ x[m + shift] with a
    zero row when (m % T) + shift < 0."""
    M = x.shape[0]
    out = torch.zeros(M, Wt.shape[1])
    t = torch.arange(M) % T
    for j, sh in enumerate(shifts):
        idx = torch.arange(M) + sh
        ok = (t + sh >= 0)
        src = torch.where(ok[:, None], x[idx.clamp(min=0)], torch.zeros(1, x.shape[1]))
        out = out + src @ Wt[j].T
    return out + (bias if bias is not None else 0)

def conv_taps(w, dil):
    """make_conv: (Cout, Cin, k) -> (k, Cout, Cin), shifts -(k-1-j)*dil"""
    k = w.shape[-1]
    return w.permute(2, 0, 1).contiguous(), [-(k - 1 - j) * dil for j in range(k)]

def strided_taps(w, r):
    """(Cout, Cin, 2r) -> 2 taps over super-rows of r rows: K index = jj*Cin + ci"""
    Co, Ci, k = w.shape
    assert k == 2 * r
    A = w[:, :, :r].permute(0, 2, 1).reshape(Co, r * Ci)
    Bm = w[:, :, r:].permute(0, 2, 1).reshape(Co, r * Ci)
    return torch.stack([A, Bm]), [-1, 0]

def encode(w, c, wav):
    """wav (B, samples) -> codes (B, nq, frames), channel-last staging exactly as the planned C++ orchestration."""
    B, L = wav.shape
    # first conv, Cin = 1, k=7: direct
    w0, b0 = _t(w, "encoder.layers.0.conv.weight"), _t(w, "encoder.layers.0.conv.bias")
    xp = F.pad(wav, (6, 0))
    cols = torch.stack([xp[:, j:j + L] for j in range(7)], dim=-1)          # (B, L, 7)
    x = (cols @ w0[:, 0, :].T + b0).reshape(B * L, -1)                       # (B*L, 64)
    T, C = L, w0.shape[0]
    idx = 1
    for ratio in reversed(c.upsampling_ratios):
        for j in range(c.num_residual_layers):
            p = f"encoder.layers.{idx}."
            Wt, sh = conv_taps(_t(w, p + "block.1.conv.weight"), c.dilation_growth_rate ** j)
            h = tap_gemm(F.elu(x), T, Wt, _t(w, p + "block.1.conv.bias"), sh)
            Wt, sh = conv_taps(_t(w, p + "block.3.conv.weight"), 1)
            x = x + tap_gemm(F.elu(h), T, Wt, _t(w, p + "block.3.conv.bias"), sh)
            idx += 1
        idx += 1
        p = f"encoder.layers.{idx}."
        a = F.elu(x).reshape(B, T, C)
        extra = (-T) % ratio
        if extra:
            a = F.pad(a, (0, 0, 0, extra))                                   # zero rows on the right
        Ts = (T + extra) // ratio
        sup = a.reshape(B * Ts, ratio * C)                                   # super-rows
        Wt, sh = strided_taps(_t(w, p + "conv.weight"), ratio)
        x = tap_gemm(sup, Ts, Wt, _t(w, p + "conv.bias"), sh)
        T, C = Ts, C * 2
        idx += 1
    idx += 1
    p = f"encoder.layers.{idx}."
    Wt, sh = conv_taps(_t(w, p + "conv.weight"), 1)
    x = tap_gemm(F.elu(x), T, Wt, _t(w, p + "conv.bias"), sh)               # (B*T, hidden)
    # transformer via the oracle's whole-sequence function (LN + attn_rows(window) + GELU MLP are per-op identical)
    x = E.encoder_transformer(w, c, x.reshape(B, T, -1)).reshape(B * T, -1)
    H = x.shape[-1]
    # downsample k=4 stride 2, replicate pad: padded = [x0, x0 | x | replicate right to even]
    a = x.reshape(B, T, H)
    extra = (-T) % 2
    left = a[:, :1].expand(B, 2, H)
    right = a[:, -1:].expand(B, extra, H)
    pad = torch.cat([left, a, right], dim=1)                                 # (B, 2 + T + extra, H)
    Ts = pad.shape[1] // 2
    sup = pad.reshape(B * Ts, 2 * H)
    Wt, sh = strided_taps(_t(w, "downsample.conv.weight"), 2)
    y = tap_gemm(sup, Ts, Wt, None, sh).reshape(B, Ts, H)[:, 1:]            # skip 1 super-row
    T = Ts - 1
    x = y.reshape(B * T, H)
    # RVQ: scores by GEMM, argmin of ||e||^2 - 2 r.e, residual update
    def rvq(prefix, n_layers):
        r = x @ _t(w, prefix + "input_proj.weight")[:, :, 0].T
        out = []
        for i in range(n_layers):
            es = _t(w, f"{prefix}layers.{i}.codebook.embed_sum"); cu = _t(w, f"{prefix}layers.{i}.codebook.cluster_usage")
            tab = es / cu.clamp(min=1e-5)[:, None]
            d = (tab * tab).sum(1)[None, :] - 2.0 * (r @ tab.T)
            ind = d.argmin(dim=-1)
            r = r - tab[ind]
            out.append(ind.reshape(B, T))
        return out
    ns = c.num_semantic_quantizers
    nq = c.encoder_valid_num_quantizers
    codes = rvq("quantizer.semantic_residual_vector_quantizer.", ns) + rvq("quantizer.acoustic_residual_vector_quantizer.", nq - ns)
    return torch.stack(codes, dim=1)
