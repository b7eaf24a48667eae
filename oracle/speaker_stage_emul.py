"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Python mirror of `qtts_speaker::embed` (qwen3-tts_amd/csrc/speaker_engine.hip): the STFT as a 4-tap GEMM over rows of
`hop` samples against a Hann-windowed DFT matrix, the reflect "same" padding of the TDNN convs as a staging step whose
causal-tap GEMM output is read `(k-1)*dilation` rows later, Res2Net channel slices, squeeze-excitation and attentive
statistics pooling on channel-last rows -- checked against oracle/speaker_ref.py (pinned to the reference module) in
tests/test_oracle_golden.py.  It verifies the ALGEBRA of the C++ orchestration."""
import math

import torch
import torch.nn.functional as F

import speaker_ref as S
from speaker_ref import _t

def tap_gemm(x, T, Wt, bias, shifts):
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
    k = w.shape[-1]
    return w.permute(2, 0, 1).contiguous(), [-(k - 1 - j) * dil for j in range(k)]

def reflect_rows(x, p):            # x (B, T, C) -> (B, T+2p, C), torch 'reflect' (no edge repeat)
    if p == 0:
        return x
    T = x.shape[1]
    idx = torch.arange(-p, T + p)
    idx = torch.where(idx < 0, -idx, idx)
    idx = torch.where(idx >= T, 2 * (T - 1) - idx, idx)
    return x[:, idx]

def tdnn(w, p, x, B, T, dil, act=True):
    wt = _t(w, p + "conv.weight"); b = _t(w, p + "conv.bias")
    k = wt.shape[-1]; total = dil * (k - 1); pad = total // 2
    xp = reflect_rows(x.reshape(B, T, -1), pad)
    Wt, sh = conv_taps(wt, dil)
    y = tap_gemm(xp.reshape(B * (T + total), -1), T + total, Wt, b, sh).reshape(B, T + total, -1)[:, total:]
    y = y.reshape(B * T, -1)
    return F.relu(y) if act else y

def speaker(w, c, mels):           # mels (B, T, mel_dim)
    B, T, _ = mels.shape
    ch, ks, dil = list(c.enc_channels), list(c.enc_kernel_sizes), list(c.enc_dilations)
    h = tdnn(w, "blocks.0.", mels.reshape(B * T, -1), B, T, dil[0])
    feats = []
    for i in range(1, len(ch) - 1):
        p = f"blocks.{i}."
        r = h
        h1 = tdnn(w, p + "tdnn1.", h, B, T, 1)
        sc = c.enc_res2net_scale; cw = ch[i] // sc
        outs = []; prev = None
        for s in range(sc):
            part = h1[:, s * cw:(s + 1) * cw]
            if s == 0: o = part
            elif s == 1: o = tdnn(w, f"{p}res2net_block.blocks.{s - 1}.", part, B, T, dil[i])
            else: o = tdnn(w, f"{p}res2net_block.blocks.{s - 1}.", part + prev, B, T, dil[i])
            outs.append(o); prev = o
        h2 = torch.cat(outs, dim=1)
        h3 = tdnn(w, p + "tdnn2.", h2, B, T, 1)
        m = h3.reshape(B, T, -1).mean(1)                                     # (B, C)
        g = F.relu(m @ _t(w, p + "se_block.conv1.weight")[:, :, 0].T + _t(w, p + "se_block.conv1.bias"))
        g = torch.sigmoid(g @ _t(w, p + "se_block.conv2.weight")[:, :, 0].T + _t(w, p + "se_block.conv2.bias"))
        h = (h3.reshape(B, T, -1) * g[:, None, :]).reshape(B * T, -1) + r
        feats.append(h)
    x = tdnn(w, "mfa.", torch.cat(feats, dim=1), B, T, dil[-1])              # (B*T, C3)
    C3 = x.shape[1]
    xb = x.reshape(B, T, C3)
    def stats(wgt):                                                           # wgt (B, T, C3)
        mean = (wgt * xb).sum(1)
        std = torch.sqrt((wgt * (xb - mean[:, None]) ** 2).sum(1).clamp(1e-12))
        return mean, std
    mean, std = stats(torch.full((B, T, C3), 1.0 / T))
    att_in = torch.cat([xb, mean[:, None].expand(B, T, C3), std[:, None].expand(B, T, C3)], dim=2).reshape(B * T, 3 * C3)
    a = torch.tanh(tdnn(w, "asp.tdnn.", att_in, B, T, 1))
    a = a @ _t(w, "asp.conv.weight")[:, :, 0].T + _t(w, "asp.conv.bias")
    a = torch.softmax(a.reshape(B, T, C3), dim=1)
    mean, std = stats(a)
    pooled = torch.cat([mean, std], dim=1)                                    # (B, 2*C3)
    return pooled @ _t(w, "fc.weight")[:, :, 0].T + _t(w, "fc.bias")

def mel_gemm(wav, n_fft=1024, hop=256, n_mels=128, sr=24000, fmin=0, fmax=12000):
    B, Sn = wav.shape
    pad = (n_fft - hop) // 2
    idx = torch.arange(-pad, Sn + pad)
    idx = torch.where(idx < 0, -idx, idx); idx = torch.where(idx >= Sn, 2 * (Sn - 1) - idx, idx)
    xp = wav[:, idx]
    R = xp.shape[1] // hop
    rows = xp[:, :R * hop].reshape(B * R, hop)
    taps = n_fft // hop
    n = torch.arange(n_fft, dtype=torch.float64)
    hann = 0.5 - 0.5 * torch.cos(2 * math.pi * n / n_fft)                     # periodic Hann (torch.hann_window default)
    f = torch.arange(n_fft // 2 + 1, dtype=torch.float64)
    ang = 2 * math.pi * f[:, None] * n[None, :] / n_fft
    Wre = (hann[None, :] * torch.cos(ang)); Wim = (-hann[None, :] * torch.sin(ang))
    W = torch.cat([Wre, Wim], dim=0).float()                                  # (1026, 1024)
    Wt = W.reshape(W.shape[0], taps, hop).permute(1, 0, 2).contiguous()       # (taps, 1026, 256)
    y = tap_gemm(rows, R, Wt, None, [-(taps - 1 - j) for j in range(taps)]).reshape(B, R, -1)[:, taps - 1:]
    nb = n_fft // 2 + 1
    mag = torch.sqrt(y[..., :nb] ** 2 + y[..., nb:] ** 2 + 1e-9)
    fb = torch.from_numpy(S.mel_filterbank_slaney(sr, n_fft, n_mels, fmin, fmax))
    return torch.log(torch.clamp(mag @ fb.T, min=1e-5))                       # (B, L, n_mels)
