"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU fp32 restatement of the speaker branch of the Base (voice-clone) model -- SURVEY.md 8(f4), the oracle for a later
HIP implementation:

  * `speaker_encoder_forward`  Qwen3TTSSpeakerEncoder (ECAPA-TDNN), modeling_qwen3_tts.py:95-393 (M below): TDNN ->
    3 x SE-Res2Net -> multi-layer feature aggregation -> attentive statistics pooling -> 1x1 conv.  PINNED against the
    reference module itself (tests/golden/speaker_tiny.npz, oracle/gen_golden.py).
  * `mel_spectrogram`          M:399-464: reflect pad, Hann STFT (center=False), magnitude with +1e-9, Slaney mel
    filterbank, log(clamp(., 1e-5)).  The filterbank comes from `librosa.filters.mel` in the reference -- a third-party
    dependency that is absent here (librosa 0.10.x in upstream environments; not pinned by pyproject.toml) -- so
    `mel_filterbank_slaney` restates its published algorithm (Slaney mel scale, `norm="slaney"` area normalisation,
    htk=False).  **Parity of the filterbank itself is unpinned**; everything downstream of it is pinned by feeding the
    same filterbank to the reference's arithmetic (the STFT / log path is torch in both).
  * `extract_speaker_embedding`  M:1941-1954 (24 kHz only, n_fft 1024, hop 256, 128 mels, fmax 12 kHz).

Weights: flat {name: tensor} dict with the reference's state_dict names relative to `speaker_encoder.`.
"""
from typing import List

import numpy as np
import torch
import torch.nn.functional as F


def _t(w, k):
    v = w[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def _conv_same_reflect(x, weight, bias, dilation=1):
    """nn.Conv1d(padding="same", padding_mode="reflect") as used by every layer here (M:264-271): total padding
    dilation*(k-1), split left = total//2, right = total - left, reflect mode."""
    k = weight.shape[-1]
    total = dilation * (k - 1)
    left = total // 2
    if total > 0:
        x = F.pad(x, (left, total - left), mode="reflect")
    return F.conv1d(x, weight, bias, dilation=dilation)


def _tdnn(w, p, x, dilation):
    """TimeDelayNetBlock M:254-274: conv -> ReLU."""
    return F.relu(_conv_same_reflect(x, _t(w, p + "conv.weight"), _t(w, p + "conv.bias"), dilation))


def _res2net(w, p, x, scale, dilation):
    """Res2NetBlock M:95-126: split channels into `scale` groups; group 0 passes through, group i>=1 goes through
    its own TDNN after adding the previous group's output (i >= 2)."""
    outs = []
    prev = None
    for i, part in enumerate(torch.chunk(x, scale, dim=1)):
        if i == 0:
            o = part
        elif i == 1:
            o = _tdnn(w, f"{p}blocks.{i - 1}.", part, dilation)
        else:
            o = _tdnn(w, f"{p}blocks.{i - 1}.", part + prev, dilation)
        outs.append(o)
        prev = o
    return torch.cat(outs, dim=1)


def _se(w, p, x):
    """SqueezeExcitationBlock M:129-158: channel gate from the time mean."""
    m = x.mean(dim=2, keepdim=True)
    m = F.relu(F.conv1d(m, _t(w, p + "conv1.weight"), _t(w, p + "conv1.bias")))
    m = torch.sigmoid(F.conv1d(m, _t(w, p + "conv2.weight"), _t(w, p + "conv2.bias")))
    return x * m


def _asp(w, p, x):
    """AttentiveStatisticsPooling M:161-251 with full-length sequences (lengths = 1.0 * L, so the mask is all ones):
    global mean/std -> attention over time per channel -> weighted mean/std."""
    eps = 1e-12
    L = x.shape[-1]

    def stats(x_, m):
        mean = (m * x_).sum(2)
        std = torch.sqrt((m * (x_ - mean.unsqueeze(2)).pow(2)).sum(2).clamp(eps))
        return mean, std

    uni = torch.ones(x.shape[0], 1, L, dtype=x.dtype) / L
    mean, std = stats(x, uni)
    att = torch.cat([x, mean.unsqueeze(2).repeat(1, 1, L), std.unsqueeze(2).repeat(1, 1, L)], dim=1)
    att = torch.tanh(_tdnn(w, p + "tdnn.", att, 1))
    att = F.conv1d(att, _t(w, p + "conv.weight"), _t(w, p + "conv.bias"))
    att = F.softmax(att, dim=2)
    mean, std = stats(x, att)
    return torch.cat((mean, std), dim=1).unsqueeze(2)


def speaker_encoder_forward(w, cfg, mels: torch.Tensor) -> torch.Tensor:
    """Qwen3TTSSpeakerEncoder.forward M:369-393.  mels (B, L, mel_dim) -> (B, enc_dim).
    cfg: enc_channels, enc_kernel_sizes, enc_dilations, enc_res2net_scale (Qwen3TTSSpeakerEncoderConfig, C:22-67)."""
    ch, ks, dil = list(cfg.enc_channels), list(cfg.enc_kernel_sizes), list(cfg.enc_dilations)
    h = mels.transpose(1, 2)
    feats: List[torch.Tensor] = []
    h = _tdnn(w, "blocks.0.", h, dil[0])
    feats.append(h)
    for i in range(1, len(ch) - 1):
        p = f"blocks.{i}."
        r = h
        h = _tdnn(w, p + "tdnn1.", h, 1)
        h = _res2net(w, p + "res2net_block.", h, cfg.enc_res2net_scale, dil[i])
        h = _tdnn(w, p + "tdnn2.", h, 1)
        h = _se(w, p + "se_block.", h) + r                                         # M:300-311
        feats.append(h)
    h = torch.cat(feats[1:], dim=1)
    h = _tdnn(w, "mfa.", h, dil[-1])
    h = _asp(w, "asp.", h)
    h = F.conv1d(h, _t(w, "fc.weight"), _t(w, "fc.bias"))
    return h.squeeze(-1)


# ----------------------------------------------------------------------------- mel front end
def mel_filterbank_slaney(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """`librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` with its defaults htk=False, norm="slaney" (published
    algorithm: Slaney's Auditory Toolbox mel scale -- linear below 1 kHz at 200/3 Hz per mel, logarithmic above with
    step log(6.4)/27 -- triangular filters between consecutive mel points, each scaled by 2 / (f[i+2] - f[i])).
    Returns (n_mels, 1 + n_fft//2) float32.  Unpinned: librosa is not installed in this environment."""
    fmax = float(sr) / 2 if fmax is None else float(fmax)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fft_f = np.linspace(0.0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    wts = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        wts[i] = np.maximum(0.0, np.minimum(lower, upper))
    wts *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return wts.astype(np.float32)


def mel_spectrogram(y: torch.Tensor, n_fft=1024, num_mels=128, sampling_rate=24000, hop_size=256, win_size=1024, fmin=0,
                    fmax=12000, mel_basis: torch.Tensor = None) -> torch.Tensor:
    """M:402-464 (center=False).  y (B, samples) in [-1, 1] -> (B, num_mels, frames)."""
    if mel_basis is None:
        mel_basis = torch.from_numpy(mel_filterbank_slaney(sampling_rate, n_fft, num_mels, fmin, fmax))
    pad = (n_fft - hop_size) // 2
    y = F.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=torch.hann_window(win_size), center=False,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.sqrt(torch.view_as_real(spec).pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(mel_basis, spec), min=1e-5))


def extract_speaker_embedding(w, cfg, audio: np.ndarray, sr: int, mel_basis: torch.Tensor = None) -> torch.Tensor:
    """Qwen3TTSForConditionalGeneration.extract_speaker_embedding M:1941-1954 -> (enc_dim,)."""
    assert sr == 24000, "Only support 24kHz audio"
    mels = mel_spectrogram(torch.from_numpy(np.asarray(audio, dtype=np.float32)).unsqueeze(0), mel_basis=mel_basis).transpose(1, 2)
    return speaker_encoder_forward(w, cfg, mels)[0]
