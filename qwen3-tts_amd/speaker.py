"""Speaker-embedding engine of the Base (voice-clone) model, SURVEY.md 8(f4).

Host mirror of `Qwen3TTSForConditionalGeneration.extract_speaker_embedding` (modeling_qwen3_tts.py:1941-1954) over
`qtts_speaker_*` (include/qtts.h): log-mel front end + ECAPA-TDNN.  The Slaney mel filterbank the reference takes from
`librosa.filters.mel` (absent here) is computed by `mel_filterbank_slaney` below -- a restatement of librosa's published
algorithm (htk=False, norm="slaney"), pinned against the values librosa's documentation publishes (round 4; CPU test).
STATUS round 1: HIP side compiled, orchestration executed on CPU stand-ins (tests/test_hostemu.py), hardware run pending.
"""
import ctypes as C
import threading
from dataclasses import dataclass
from typing import Any, Dict, Tuple

import numpy as np
import torch

from . import _lib
from .config import _get, _pick


@dataclass
class SpeakerEncoderConfig:
    """Qwen3TTSSpeakerEncoderConfig (configuration_qwen3_tts.py:22-67) + the mel constants of M:1944-1952."""
    mel_dim: int = 128
    enc_dim: int = 1024
    enc_channels: Tuple[int, ...] = (512, 512, 512, 512, 1536)
    enc_kernel_sizes: Tuple[int, ...] = (5, 3, 3, 3, 1)
    enc_dilations: Tuple[int, ...] = (1, 2, 3, 4, 1)
    enc_attention_channels: int = 128
    enc_res2net_scale: int = 8
    enc_se_channels: int = 128
    sample_rate: int = 24000
    n_fft: int = 1024
    hop_size: int = 256
    win_size: int = 1024
    num_mels: int = 128
    fmin: float = 0.0
    fmax: float = 12000.0

    @classmethod
    def from_any(cls, src: Any) -> "SpeakerEncoderConfig":
        if isinstance(src, cls):
            return src
        sub = _get(src, "speaker_encoder_config", None)
        kw = _pick(cls, sub if sub is not None else src)
        for k in ("enc_channels", "enc_kernel_sizes", "enc_dilations"):
            if k in kw:
                kw[k] = tuple(int(x) for x in kw[k])
        return cls(**kw)


_F_SP, _MIN_LOG_HZ = 200.0 / 3, 1000.0
_MIN_LOG_MEL, _LOGSTEP = _MIN_LOG_HZ / _F_SP, np.log(6.4) / 27.0


def hz_to_mel_slaney(f):
    """`librosa.hz_to_mel(f, htk=False)`: linear below 1 kHz (200/3 Hz per mel), logarithmic above (27 mels per factor 6.4)."""
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= _MIN_LOG_HZ, _MIN_LOG_MEL + np.log(np.maximum(f, 1e-10) / _MIN_LOG_HZ) / _LOGSTEP, f / _F_SP)


def mel_to_hz_slaney(m):
    """`librosa.mel_to_hz(m, htk=False)`."""
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= _MIN_LOG_MEL, _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL)), _F_SP * m)


def mel_frequencies_slaney(n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """`librosa.mel_frequencies(n_mels, fmin=fmin, fmax=fmax, htk=False)`: n_mels points evenly spaced on the Slaney mel scale."""
    return mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels))


def mel_filterbank_slaney(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """`librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` with its defaults (htk=False, norm="slaney"), the filterbank of the
    reference's `mel_spectrogram` (M:399-464): triangular filters between consecutive points of `mel_frequencies(n_mels + 2)`, each
    scaled by 2 / (f[i+2] - f[i]).  Returns (n_mels, 1 + n_fft//2) float32.  librosa is not in this image; the scale and the
    filterbank are pinned against the values librosa's own documentation publishes (tests/test_host_logic.py:
    `test_slaney_mel_filterbank_matches_librosa_published_values`)."""
    fmax = float(sr) / 2 if fmax is None else float(fmax)
    fft_f = np.linspace(0.0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = mel_frequencies_slaney(n_mels + 2, fmin, fmax)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    wts = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        wts[i] = np.maximum(0.0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    wts *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return wts.astype(np.float32)


class SpeakerEncoderEngine:
    """Owns one `qtts_speaker` handle."""

    def __init__(self, config: Any, state_dict: Dict[str, torch.Tensor], compute_dtype: torch.dtype = torch.float32,
                 device: str = "cuda:0", max_batch: int = 4, max_samples: int = 24000 * 30):
        self.config = SpeakerEncoderConfig.from_any(config)
        self.device = _lib.hip_device(device, "SpeakerEncoderEngine")
        self.max_batch, self.max_samples = int(max_batch), int(max_samples)
        self._lib = _lib.load_library()
        self._lock = threading.RLock()
        c = self.config
        sc = fill_speaker_config(c, compute_dtype, self.max_batch, self.max_samples)
        self._h = C.c_void_p()
        sd = {k[len("speaker_encoder."):]: v for k, v in state_dict.items() if k.startswith("speaker_encoder.")} or state_dict
        with torch.cuda.device(self.device):
            _lib.check(self._lib.qtts_speaker_create(C.byref(sc), C.byref(self._h)))
            for name, t in sd.items():
                _lib.bind_tensor(self._lib.qtts_speaker_bind, self._h, name, t)
            fb = mel_filterbank_slaney(c.sample_rate, c.n_fft, c.num_mels, c.fmin, c.fmax)
            _lib.bind_tensor(self._lib.qtts_speaker_bind, self._h, "mel_basis", torch.from_numpy(fb))
            _lib.check(self._lib.qtts_speaker_finalize(self._h))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.qtts_speaker_destroy(h)
            self._h = None

    @_lib.locked
    def embed(self, wavs: torch.Tensor) -> torch.Tensor:
        """wavs (B, samples) float in [-1, 1] at 24 kHz (equal lengths) -> (B, enc_dim)."""
        if wavs.dim() != 2:
            raise ValueError(f"wavs must be (batch, samples), got {tuple(wavs.shape)}")
        B, S = wavs.shape
        if B > self.max_batch or S > self.max_samples:
            raise ValueError(f"embed: batch {B} / samples {S} exceed max_batch {self.max_batch} / max_samples {self.max_samples}")
        x = wavs.to(self.device, torch.float32).contiguous()
        out = torch.empty(B, self.config.enc_dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.qtts_speaker_embed(self._h, C.c_void_p(x.data_ptr()), B, S, C.c_void_p(out.data_ptr()), None,
                                                    C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    def embed_many(self, audios) -> list:
        """A LIST of 24 kHz waveforms -> their x-vectors, in order.  Clips of EQUAL length run as one batch of up to `max_batch` rows
        (every row keeps its own time statistics -- the squeeze-excitation means and the attentive pooling are per row -- so a row's
        x-vector does not depend on its neighbours); clips of other lengths form their own groups: padding a clip would change its
        statistics, so ragged clips are bucketed by exact length."""
        arrs = [np.asarray(a, dtype=np.float32).reshape(-1) for a in audios]
        groups: Dict[int, list] = {}
        for i, a in enumerate(arrs):
            groups.setdefault(a.shape[0], []).append(i)
        out = [None] * len(arrs)
        for idx in groups.values():
            for k in range(0, len(idx), self.max_batch):
                part = idx[k:k + self.max_batch]
                emb = self.embed(torch.from_numpy(np.stack([arrs[i] for i in part])))
                for r, i in enumerate(part):
                    out[i] = emb[r]
        return out

    def extract_speaker_embedding(self, audio: np.ndarray, sr: int) -> torch.Tensor:
        """M:1941-1954: one waveform -> (enc_dim,)."""
        assert sr == self.config.sample_rate, "Only support 24kHz audio"
        return self.embed(torch.from_numpy(np.asarray(audio, dtype=np.float32)).unsqueeze(0))[0]


def fill_speaker_config(c: SpeakerEncoderConfig, compute_dtype, max_batch: int, max_samples: int):
    sc = _lib.SpeakerConfigC()
    sc.mel_dim, sc.enc_dim, sc.n_blocks = int(c.mel_dim), int(c.enc_dim), len(c.enc_channels)
    for i in range(len(c.enc_channels)):
        sc.channels[i], sc.kernel_sizes[i], sc.dilations[i] = int(c.enc_channels[i]), int(c.enc_kernel_sizes[i]), int(c.enc_dilations[i])
    sc.attention_channels, sc.res2net_scale, sc.se_channels = int(c.enc_attention_channels), int(c.enc_res2net_scale), int(c.enc_se_channels)
    sc.n_fft, sc.hop_size, sc.win_size, sc.num_mels = int(c.n_fft), int(c.hop_size), int(c.win_size), int(c.num_mels)
    sc.compute_dtype = _lib.QTTS_BF16 if compute_dtype == torch.bfloat16 else _lib.QTTS_F32
    sc.max_batch, sc.max_samples = int(max_batch), int(max_samples)
    return sc
