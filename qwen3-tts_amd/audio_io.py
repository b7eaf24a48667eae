"""Audio inputs of the reference's wrappers: wav path / URL / base64 string -> mono float32 waveform, and resampling.

The reference does this with third-party code that is not in this image -- `soundfile.read(dtype="float32")` for bytes,
`librosa.load(path, sr=None, mono=True)` for paths and `librosa.resample` (default `res_type="soxr_hq"`) for rate
conversion (qwen_tts/inference/qwen3_tts_tokenizer.py:101-160 = IT, qwen3_tts_model.py:188-222 = IM).  This module
restates what those calls do for RIFF/WAVE files (the format the reference's examples use) on numpy + scipy:

  * `read_wav_bytes` -- PCM 8 / 16 / 24 / 32-bit and IEEE float 32 / 64, plain or WAVE_FORMAT_EXTENSIBLE, scaled to
    [-1, 1) exactly like libsndfile's float conversion (int16 / 2^15, int24 / 2^23, int32 / 2^31, (uint8 - 128) / 2^7);
    exact, covered by tests/test_host_logic.py;
  * `resample` -- band-limited polyphase resampling (`scipy.signal.resample_poly`, Kaiser beta 14.77 ~ soxr "HQ"
    stop-band).  soxr itself is not available, so RESAMPLING PARITY IS UNPINNED: the output has librosa's length
    (ceil(n * target / orig)) and agrees with an ideal band-limited resampler to ~1e-4 on in-band content, but it is not
    sample-identical to soxr.  Audio already at the target rate never passes through it.

Other containers (flac / ogg / ...) go to `soundfile.read` -- the reference's own reader (IM:207-223) -- when that package is
importable (`read_audio_bytes`); where it is not (this image), they raise ValueError as before.  Pure host code: no device work,
nothing on the hot path."""
import base64
import math
import struct
import urllib.request
from fractions import Fraction
from typing import Tuple
from urllib.parse import urlparse

import numpy as np


def is_probably_base64(s: str) -> bool:
    """IT:101-107 / IM:188-194: a data URL, or a long string without path separators."""
    if s.startswith("data:audio"):
        return True
    return ("/" not in s and "\\" not in s) and len(s) > 256


def is_url(s: str) -> bool:
    """IT:109-114 / IM:196-200."""
    try:
        u = urlparse(s)
        return u.scheme in ("http", "https") and bool(u.netloc)
    except Exception:
        return False


def decode_base64_to_wav_bytes(b64: str) -> bytes:
    """IT:116-120: both 'data:audio/wav;base64,....' and raw base64."""
    if "," in b64 and b64.strip().startswith("data:"):
        b64 = b64.split(",", 1)[1]
    return base64.b64decode(b64)


_PCM, _FLOAT, _EXTENSIBLE = 0x0001, 0x0003, 0xFFFE


def read_wav_bytes(data: bytes) -> Tuple[np.ndarray, int]:
    """RIFF/WAVE bytes -> (float32 array of shape (n,) for mono or (n, channels), sample rate): the result of
    `soundfile.read(io.BytesIO(data), dtype="float32", always_2d=False)` for PCM / IEEE-float WAVE files."""
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("unsupported audio container: only RIFF/WAVE is readable without libsndfile")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            if size < 16:
                raise ValueError("corrupt WAVE file: short fmt chunk")
            tag, channels, rate, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == _EXTENSIBLE and size >= 26:
                tag = struct.unpack("<H", body[24:26])[0]                 # first two bytes of the sub-format GUID
            fmt = (tag, channels, rate, bits)
        elif cid == b"data":
            payload = body                                               # a truncated data chunk yields what is there
        pos += 8 + size + (size & 1)                                     # chunks are word-aligned
    if fmt is None or payload is None:
        raise ValueError("corrupt WAVE file: missing fmt or data chunk")
    tag, channels, rate, bits = fmt
    if channels < 1:
        raise ValueError("corrupt WAVE file: zero channels")
    if tag == _PCM and bits == 8:
        x = (np.frombuffer(payload, np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == _PCM and bits == 16:
        x = np.frombuffer(payload[:len(payload) // 2 * 2], "<i2").astype(np.float32) / 32768.0
    elif tag == _PCM and bits == 24:
        b = np.frombuffer(payload[:len(payload) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) / 8388608.0
    elif tag == _PCM and bits == 32:
        x = (np.frombuffer(payload[:len(payload) // 4 * 4], "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == _FLOAT and bits == 32:
        x = np.frombuffer(payload[:len(payload) // 4 * 4], "<f4").astype(np.float32)
    elif tag == _FLOAT and bits == 64:
        x = np.frombuffer(payload[:len(payload) // 8 * 8], "<f8").astype(np.float32)
    else:
        raise ValueError(f"unsupported WAVE encoding: format tag {tag:#06x}, {bits} bits")
    x = x[:len(x) // channels * channels]
    return (x if channels == 1 else x.reshape(-1, channels)), int(rate)


def read_audio_bytes(data: bytes) -> Tuple[np.ndarray, int]:
    """Any container the reference reads (IM:207-223: `soundfile.read(io.BytesIO(bytes), dtype="float32", always_2d=False)`):
    RIFF/WAVE by this module's own exact reader; everything else (FLAC, OGG/Vorbis, AIFF, ...) by `soundfile` when it is importable;
    without it, the ValueError `read_wav_bytes` raises for a non-WAVE container."""
    if len(data) >= 12 and data[:4] == b"RIFF" and data[8:12] == b"WAVE":
        return read_wav_bytes(data)
    try:
        import soundfile
    except ImportError:
        return read_wav_bytes(data)                                      # raises: "only RIFF/WAVE is readable without libsndfile"
    import io
    audio, sr = soundfile.read(io.BytesIO(data), dtype="float32", always_2d=False)
    return np.asarray(audio, dtype=np.float32), int(sr)


def load_audio_to_np(x: str) -> Tuple[np.ndarray, int]:
    """IM:207-222: path / URL / base64 -> (mono float32 waveform, its own sample rate)."""
    if is_url(x):
        with urllib.request.urlopen(x) as resp:
            audio, sr = read_audio_bytes(resp.read())
    elif is_probably_base64(x):
        audio, sr = read_audio_bytes(decode_base64_to_wav_bytes(x))
    else:
        with open(x, "rb") as f:                                         # FileNotFoundError like librosa.load
            audio, sr = read_audio_bytes(f.read())
    if audio.ndim > 1:
        audio = np.mean(audio, axis=-1)
    return audio.astype(np.float32), int(sr)


def resample(y: np.ndarray, orig_sr: int, target_sr: int) -> np.ndarray:
    """`librosa.resample(y=y, orig_sr=orig_sr, target_sr=target_sr)` for 1-D input: same output length
    (ceil(n * target / orig)), band-limited polyphase kernel (see the module docstring: parity with soxr unpinned)."""
    y = np.asarray(y, dtype=np.float32)
    if y.ndim != 1:
        raise ValueError("resample expects a 1-D waveform")
    orig_sr, target_sr = int(orig_sr), int(target_sr)
    if orig_sr <= 0 or target_sr <= 0:
        raise ValueError("sample rates must be positive")
    if orig_sr == target_sr:
        return y
    n_out = int(math.ceil(len(y) * target_sr / orig_sr))
    if len(y) == 0:
        return np.zeros(0, np.float32)
    from scipy.signal import firwin, resample_poly
    r = Fraction(target_sr, orig_sr)
    up, down = r.numerator, r.denominator
    # anti-alias / anti-image low-pass at 0.95 x the lower Nyquist: 32 zero crossings per side at the lower rate
    m = max(up, down)
    taps = firwin(2 * 32 * m + 1, 0.95 / m, window=("kaiser", 14.769656459379492)).astype(np.float64)
    out = resample_poly(y.astype(np.float64), up, down, window=taps)
    if len(out) < n_out:
        out = np.pad(out, (0, n_out - len(out)))
    return out[:n_out].astype(np.float32)


def load_audio(x: str, target_sr: int) -> np.ndarray:
    """IT:122-160: load, down-mix, resample to `target_sr`."""
    audio, sr = load_audio_to_np(x)
    if sr != int(target_sr):
        audio = resample(audio, sr, int(target_sr))
    return audio.astype(np.float32)
