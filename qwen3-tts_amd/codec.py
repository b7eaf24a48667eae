"""Host side of the 12 Hz codec decoder: mirrors of the reference's decode path over libqtts.

Reference classes mirrored (same names, argument meaning and error behaviour):
  * Qwen3TTSTokenizerV2Decoder.forward / .chunked_decode   tokenizer_12hz/modeling_..._v2.py:869-896
  * Qwen3TTSTokenizerV2Model.decode                        tokenizer_12hz/modeling_..._v2.py:993-1024
  * Qwen3TTSTokenizer.decode (input normalisation)         qwen_tts/inference/qwen3_tts_tokenizer.py:259-365
All arithmetic runs in the HIP library; torch only owns the device buffers.
"""
import ctypes as C
import threading
import json
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

from . import _lib
from .config import CodecDecoderConfig


def _default_inv_freq(theta: float, head_dim: int) -> torch.Tensor:
    """HF 'default' rope init: 1 / theta^(2i/d), computed by torch exactly as the reference's
    rotary module does (tokenizer v2:260-263 -> ROPE_INIT_FUNCTIONS['default'])."""
    return 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(dtype=torch.float) / head_dim))


def _relative_decoder_state(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept either the full tokenizer checkpoint (`decoder.*` + `encoder.*`) or the decoder's own
    state_dict; return names relative to the decoder module."""
    if any(k.startswith("decoder.quantizer.") for k in sd):
        return {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    return dict(sd)


class CodecDecoderEngine:
    """Owns one `qtts_codec` handle (Qwen3TTSTokenizerV2Decoder on the GPU)."""

    def __init__(self, config: Any, state_dict: Dict[str, torch.Tensor], compute_dtype: torch.dtype = torch.float32,
                 device: str = "cuda:0", max_batch: int = 8, max_frames: int = 325):
        self.config = CodecDecoderConfig.from_any(config)
        self.device = _lib.hip_device(device, "CodecDecoderEngine")
        self.compute_dtype = compute_dtype
        self.max_batch, self.max_frames = int(max_batch), int(max_frames)
        self._lib = _lib.load_library()
        self._lock = threading.RLock()
        c = self.config
        cc = _lib.CodecConfigC()
        for f in ("codebook_size", "codebook_dim", "hidden_size", "latent_dim", "num_attention_heads",
                  "num_key_value_heads", "head_dim", "sliding_window", "intermediate_size", "num_hidden_layers",
                  "num_quantizers", "decoder_dim"):
            setattr(cc, f, int(getattr(c, f)))
        cc.n_upsample_rates = len(c.upsample_rates)
        cc.n_upsampling_ratios = len(c.upsampling_ratios)
        for i, r in enumerate(c.upsample_rates):
            cc.upsample_rates[i] = int(r)
        for i, r in enumerate(c.upsampling_ratios):
            cc.upsampling_ratios[i] = int(r)
        cc.rms_norm_eps, cc.rope_theta = float(c.rms_norm_eps), float(c.rope_theta)
        cc.compute_dtype = _lib.QTTS_BF16 if compute_dtype == torch.bfloat16 else _lib.QTTS_F32
        cc.max_batch, cc.max_frames = self.max_batch, self.max_frames
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.qtts_codec_create(C.byref(cc), C.byref(self._h)))
            sd = _relative_decoder_state(state_dict)
            for name, t in sd.items():
                if ".input_proj." in name and name.startswith("quantizer."):
                    continue  # encode-side projection, unused by decode (v2:758-760)
                _lib.bind_tensor(self._lib.qtts_codec_bind, self._h, name, t)
            _lib.bind_tensor(self._lib.qtts_codec_bind, self._h, "pre_transformer.rotary_emb.inv_freq",
                             _default_inv_freq(c.rope_theta, c.head_dim))
            _lib.check(self._lib.qtts_codec_finalize(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.qtts_codec_destroy(h)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _call(rc: int):
        """`_lib.check`, with the library's "code index out of range" mapped to the IndexError the reference's embedding
        lookup raises for a code >= codebook_size (v2:721-727)."""
        try:
            _lib.check(rc)
        except _lib.QttsError as e:
            if "code index out of range" in str(e):
                raise IndexError(str(e)) from e
            raise

    def _check_codes(self, codes: torch.Tensor, q_dim: int):
        if codes.dim() != 3:
            raise ValueError(f"codes must be 3-D, got shape {tuple(codes.shape)}")
        if codes.shape[q_dim] != self.config.num_quantizers:
            raise ValueError(f"Expected {self.config.num_quantizers} layer of codes, got {codes.shape[q_dim]}")  # v2:870-871

    @_lib.locked
    def forward(self, codes: torch.Tensor, return_pre_clamp: bool = False):
        """Qwen3TTSTokenizerV2Decoder.forward (v2:869-884): codes (B, Q, T) -> (B, 1, T*1920)."""
        self._check_codes(codes, 1)
        codes = codes.to(self.device, torch.int64).contiguous()
        B, _, T = codes.shape
        up = self.config.total_upsample
        wav = torch.empty(B, T * up, dtype=torch.float32, device=self.device)
        pre = torch.empty_like(wav) if return_pre_clamp else None
        with torch.cuda.device(self.device):
            self._call(self._lib.qtts_codec_forward(self._h, C.c_void_p(codes.data_ptr()), B, T, C.c_void_p(wav.data_ptr()),
                                                    C.c_void_p(pre.data_ptr()) if pre is not None else None, self._stream()))
        return (wav.unsqueeze(1), pre.unsqueeze(1)) if return_pre_clamp else wav.unsqueeze(1)

    __call__ = forward

    @_lib.locked
    def forward_stage(self, codes: torch.Tensor, stage: str) -> torch.Tensor:
        """Diagnostic: activation after `stage`, channel-last (B, L, C)."""
        self._check_codes(codes, 1)
        codes = codes.to(self.device, torch.int64).contiguous()
        B, _, T = codes.shape
        cap = B * T * self.config.total_upsample * max(self.config.decoder_dim, 4 * self.config.latent_dim)
        out = torch.empty(cap, dtype=torch.float32, device=self.device)
        L, Cc = C.c_int64(), C.c_int64()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.qtts_codec_forward_stage(self._h, C.c_void_p(codes.data_ptr()), B, T, stage.encode(),
                                                          C.c_void_p(out.data_ptr()), cap, C.byref(L), C.byref(Cc), self._stream()))
        return out[: B * L.value * Cc.value].view(B, L.value, Cc.value)

    @_lib.locked
    def decode_padded(self, audio_codes: torch.Tensor, chunk_size: int = 300, left_context_size: int = 25
                      ) -> Tuple[torch.Tensor, List[int]]:
        """Body of Qwen3TTSTokenizerV2Model.decode (v2:1012-1015): audio_codes (B, T, Q) padded with -1
        -> (wav (B, T*1920), lengths)."""
        self._check_codes(audio_codes, 2)
        codes = audio_codes.to(self.device, torch.int64).contiguous()
        B, T, _ = codes.shape
        up = self.config.total_upsample
        wav = torch.empty(B, T * up, dtype=torch.float32, device=self.device)
        lens = (C.c_int64 * B)()
        with torch.cuda.device(self.device):
            self._call(self._lib.qtts_codec_decode(self._h, C.c_void_p(codes.data_ptr()), B, T, int(chunk_size),
                                                   int(left_context_size), C.c_void_p(wav.data_ptr()), lens, self._stream()))
        return wav, [int(x) for x in lens]

    def stats(self) -> Dict[str, int]:
        """Graph-cache bookkeeping of this engine (include/qtts.h `qtts_codec_get_stats`)."""
        st = _lib.CodecStatsC()
        _lib.check(self._lib.qtts_codec_get_stats(self._h, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in _lib.CodecStatsC._fields_}

    @_lib.locked
    def stream_begin(self, batch: int):
        """Start a state-carrying streaming session for `batch` sequences (include/qtts.h `qtts_codec_stream_begin`).
        Validated on MI355X in round 2 (stream == forward); `stream()` below is the stateless packet API."""
        with torch.cuda.device(self.device):
            _lib.check(self._lib.qtts_codec_stream_begin(self._h, int(batch)))
        self._stream_batch = int(batch)

    @_lib.locked
    def stream_push(self, codes: torch.Tensor) -> torch.Tensor:
        """Decode the next packet: codes (B, Q, k) int64 -> (B, 1, k * total_upsample); the handle carries the conv /
        attention state, so the concatenated packets equal `forward` on the whole sequence."""
        self._check_codes(codes, 1)
        B, _, k = codes.shape
        if getattr(self, "_stream_batch", 0) != B:
            raise ValueError("stream_push: call stream_begin(batch) with this batch size first")
        if int(codes.min()) < 0 or int(codes.max()) >= self.config.codebook_size:
            raise ValueError("stream_push: code index out of range")
        codes = codes.to(self.device, torch.long).contiguous()
        wav = torch.empty(B, k * self.config.total_upsample, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.qtts_codec_stream_push(self._h, C.c_void_p(codes.data_ptr()), int(k),
                                                        C.c_void_p(wav.data_ptr()), self._stream()))
        return wav.unsqueeze(1)

    def stream(self, left_context_size: int = 25) -> "CodecStreamDecoder":
        """A packet-by-packet decoder bound to this engine (see CodecStreamDecoder)."""
        return CodecStreamDecoder(self.forward, self.config.total_upsample, left_context_size)

    def chunked_decode(self, codes: torch.Tensor, chunk_size: int = 300, left_context_size: int = 25) -> torch.Tensor:
        """Qwen3TTSTokenizerV2Decoder.chunked_decode (v2:886-896): codes (B, Q, T) -> (B, 1, T*1920)."""
        self._check_codes(codes, 1)
        wav, _ = self.decode_padded(codes.transpose(1, 2), chunk_size, left_context_size)
        return wav.unsqueeze(1)


class CodecStreamDecoder:
    """Packet-by-packet decode for streaming output (first-packet latency, BASELINE config 4).

    The reference has no streaming OUTPUT API (qwen3_tts_model.py:513-515), but its decoder defines how a long code
    sequence is cut: `chunked_decode(codes, chunk_size, left_context_size)` (tokenizer v2:886-896) decodes every chunk
    together with up to `left_context_size` previous frames and drops the context's samples.  Pushing packets of k
    frames through this class emits exactly `chunked_decode(all_codes, chunk_size=k, left_context_size=L)`, chunk by
    chunk, while holding only the last L frames of codes.  `forward` is any `(B, Q, T) int64 -> (B, 1, T*upsample)`
    callable: `CodecDecoderEngine.forward` in the product, the oracle's decoder in the CPU test."""

    def __init__(self, forward, total_upsample: int, left_context_size: int = 25):
        self._forward = forward
        self.total_upsample = int(total_upsample)
        self.left_context_size = int(left_context_size)
        self.reset()

    def reset(self):
        self._ctx = None            # (B, Q, <= L) codes kept from earlier packets
        self._start = 0             # frames emitted so far (`start_index` of v2:888)

    def push(self, codes: torch.Tensor) -> torch.Tensor:
        """codes (B, Q, k) int64 -> waveform (B, 1, k * total_upsample) for exactly these k frames."""
        if codes.dim() != 3 or codes.shape[-1] < 1:
            raise ValueError(f"Expected codes of shape (B, Q, k >= 1), got {tuple(codes.shape)}")
        L = self.left_context_size
        ctx = L if self._start - L > 0 else self._start                        # v2:891
        chunk = codes if ctx == 0 else torch.cat([self._ctx[..., self._ctx.shape[-1] - ctx:].to(codes.device), codes], dim=-1)
        wav = self._forward(chunk)
        out = wav[..., ctx * self.total_upsample:]
        keep = chunk[..., max(0, chunk.shape[-1] - L):] if L > 0 else None
        self._ctx = keep
        self._start += int(codes.shape[-1])
        return out


@dataclass
class Qwen3TTSTokenizerV2EncoderOutput:
    """tokenizer v2:55-62."""
    audio_codes: List[torch.Tensor] = None


@dataclass
class Qwen3TTSTokenizerV2DecoderOutput:
    """tokenizer v2:63-72."""
    audio_values: List[torch.Tensor] = None


class Qwen3TTSTokenizerV2Model:
    """Decode-side mirror of Qwen3TTSTokenizerV2Model (v2:928-1024).  `encode` is outside the hot path
     and raises."""

    def __init__(self, config: Any, state_dict: Dict[str, torch.Tensor], device: str = "cuda:0",
                 dtype: torch.dtype = torch.float32, max_batch: int = 8, max_frames: int = 325):
        self.config = config
        self.decoder_config = CodecDecoderConfig.from_any(config)
        self.decoder = CodecDecoderEngine(self.decoder_config, state_dict, compute_dtype=dtype, device=device,
                                          max_batch=max_batch, max_frames=max_frames)
        self.device = self.decoder.device
        self.dtype = dtype
        self.input_sample_rate = self.decoder_config.input_sample_rate
        self.output_sample_rate = self.decoder_config.output_sample_rate
        self.decode_upsample_rate = self.decoder_config.decode_upsample_rate
        self.encode_downsample_rate = self.decoder_config.encode_downsample_rate
        # encoder (Mimi) weights are kept on the host until the first encode() call builds the HIP encoder
        enc = {k: v for k, v in state_dict.items() if k.startswith("encoder.")}
        self._encoder_state = enc if any(k.startswith("encoder.encoder.layers.") for k in enc) else None
        self._encoder = None

    def get_model_type(self):
        return "qwen3_tts_tokenizer_12hz"

    def get_input_sample_rate(self):
        return self.input_sample_rate

    def get_output_sample_rate(self):
        return self.output_sample_rate

    def get_encode_downsample_rate(self):
        return self.encode_downsample_rate

    def get_decode_upsample_rate(self):
        return self.decode_upsample_rate

    def encode(self, input_values: torch.Tensor, padding_mask: Optional[torch.Tensor] = None, return_dict: Optional[bool] = None):
        """tokenizer v2:961-991.  Needs the encoder weights (`encoder.*` keys of the tokenizer checkpoint); the HIP
        encoder is built on first use."""
        if not self._encoder_state and self._encoder is None:
            raise NotImplementedError("this tokenizer was built without encoder weights (`encoder.*` keys): "
                                      "codec encode is unavailable (SURVEY.md 8f3)")
        n_samples = int(input_values.shape[-1])
        if self._encoder is None or n_samples > self._encoder.max_samples:
            # built on first use; re-created with a larger workspace when a longer reference arrives (the reference has no
            # length limit).  Batch capacity follows the tokenizer's max_batch; larger batches run as waves below.
            from .encoder import CodecEncoderEngine
            sr = int(self.input_sample_rate)
            cap = max(30 * sr, -(-n_samples // (10 * sr)) * 10 * sr)
            self._encoder = None
            self._encoder = CodecEncoderEngine(self.config, self._encoder_state, compute_dtype=self.dtype, device=str(self.device),
                                               max_batch=self.decoder.max_batch, max_samples=cap)
        if padding_mask is None:
            padding_mask = torch.ones_like(input_values, dtype=torch.long)
        codes = []
        mb = self._encoder.max_batch
        for b0 in range(0, input_values.shape[0], mb):
            codes += self._encoder.encode(input_values[b0:b0 + mb], padding_mask[b0:b0 + mb])
        if return_dict is False:
            return (codes,)
        return Qwen3TTSTokenizerV2EncoderOutput(codes)

    def decode(self, audio_codes: torch.Tensor, return_dict: Optional[bool] = None):
        """v2:993-1024: audio_codes (B, T, Q) int64 padded with -1 -> list of 1-D waveforms."""
        B = audio_codes.shape[0]
        out: List[torch.Tensor] = []
        mb = self.decoder.max_batch
        for b0 in range(0, B, mb):      # the engine's workspace is sized for max_batch rows per call
            wav, lens = self.decoder.decode_padded(audio_codes[b0:b0 + mb])
            out += [w[:l] for w, l in zip(wav, lens)]
        if return_dict is False:
            return (out,)
        return Qwen3TTSTokenizerV2DecoderOutput(out)


class Qwen3TTSTokenizer:
    """Mirror of qwen_tts.inference.Qwen3TTSTokenizer (qwen3_tts_tokenizer.py:44-410), 12 Hz decode side."""

    def __init__(self):
        self.model = None
        self.feature_extractor = None
        self.config = None
        self.device = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, **kwargs) -> "Qwen3TTSTokenizer":
        """Loads `config.json` + `*.safetensors` of a Qwen3-TTS-Tokenizer-12Hz directory.  kwargs follow the
        reference (`device_map`, `dtype`, `attn_implementation` accepted and ignored)."""
        from safetensors.torch import load_file
        from .model import resolve_checkpoint_dir
        path = resolve_checkpoint_dir(pretrained_model_name_or_path, **kwargs)      # local directory, or a hub id (IT:62-99)
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        sd = {}
        for fn in sorted(os.listdir(path)):
            if fn.endswith(".safetensors"):
                sd.update(load_file(os.path.join(path, fn)))
        device = kwargs.get("device_map", kwargs.get("device", "cuda:0"))
        dtype = kwargs.get("dtype", kwargs.get("torch_dtype", torch.float32))
        return cls.from_state_dict(cfg, sd, device=str(device), dtype=dtype,
                                   max_batch=kwargs.get("max_batch", 8), max_frames=kwargs.get("max_frames", 325))

    @classmethod
    def from_state_dict(cls, config: Any, state_dict: Dict[str, torch.Tensor], device: str = "cuda:0",
                        dtype: torch.dtype = torch.float32, max_batch: int = 8, max_frames: int = 325):
        inst = cls()
        inst.model = Qwen3TTSTokenizerV2Model(config, state_dict, device=device, dtype=dtype, max_batch=max_batch,
                                              max_frames=max_frames)
        inst.config = config
        inst.device = inst.model.device
        return inst

    # ---- audio inputs (qwen3_tts_tokenizer.py:113-160): same helper names; the work is in audio_io.py
    def _is_probably_base64(self, s: str) -> bool:
        from . import audio_io
        return audio_io.is_probably_base64(s)

    def _is_url(self, s: str) -> bool:
        from . import audio_io
        return audio_io.is_url(s)

    def _decode_base64_to_wav_bytes(self, b64: str) -> bytes:
        from . import audio_io
        return audio_io.decode_base64_to_wav_bytes(b64)

    def load_audio(self, x: str, target_sr: int) -> np.ndarray:
        """qwen3_tts_tokenizer.py:136-160: wav path / URL / base64 -> mono float32 waveform at `target_sr`."""
        from . import audio_io
        return audio_io.load_audio(x, target_sr=target_sr)

    def _normalize_audio_inputs(self, audios, sr: Optional[int]) -> List[np.ndarray]:
        """qwen3_tts_tokenizer.py:162-206: str (wav path / URL / base64) or waveform, or a list of either -> mono float32
        waveforms at the model's input rate.  WAVE decoding and resampling are restated in audio_io.py (soundfile /
        librosa are not in this image; resampling is band-limited polyphase, not soxr -- see that module)."""
        from . import audio_io
        target_sr = int(self.model.input_sample_rate)
        if isinstance(audios, (str, np.ndarray, torch.Tensor)):
            audios = [audios]
        if len(audios) == 0:
            return []
        if isinstance(audios[0], str):
            return [self.load_audio(x, target_sr=target_sr) for x in audios]
        if sr is None:
            raise ValueError("For numpy waveform input, you must provide `sr` (original sampling rate).")    # IT:190
        out = []
        for a in audios:
            if isinstance(a, torch.Tensor):
                a = a.detach().cpu().numpy()
            if not isinstance(a, np.ndarray):
                raise TypeError("Mixed input types are not supported. Use all paths/base64 or all numpy arrays.")   # IT:196
            if a.ndim > 1:
                a = np.mean(a, axis=-1)
            if int(sr) != target_sr:
                a = audio_io.resample(a.astype(np.float32), int(sr), target_sr)
            out.append(a.astype(np.float32))
        return out

    def encode(self, audios, sr: Optional[int] = None, return_dict: bool = True):
        """qwen3_tts_tokenizer.py:208-257: wav path(s) / base64 string(s) / waveform(s) (+ `sr`) -> codes."""
        wavs = self._normalize_audio_inputs(audios, sr)
        if not wavs:
            raise ValueError("encode(): no audio given")
        L = max(w.shape[0] for w in wavs)
        # EncodecFeatureExtractor semantics: right zero padding + mask.  Built with numpy, not torch: torch's CPU kernels go parallel above 32 K
        # elements, and an OpenMP team sized for every core the box SHOWS, inside a container whose CPU quota is a fraction of them, gets
        # the whole process throttled for the rest of the scheduler period -- 85-100 ms stalls in one call of three on the MI355X box
        # (profiles/r05_voice_clone_prompt.md).  Host-side preparation on the request path stays single-threaded.
        x_np = np.zeros((len(wavs), L), np.float32)
        m_np = np.zeros((len(wavs), L), np.int64)
        for i, w in enumerate(wavs):
            x_np[i, : w.shape[0]] = w
            m_np[i, : w.shape[0]] = 1
        x, m = torch.from_numpy(x_np), torch.from_numpy(m_np)
        return self.model.encode(x, m, return_dict=return_dict)

    def decode(self, encoded) -> Tuple[List[np.ndarray], int]:
        """qwen3_tts_tokenizer.py:259-365 for the 12 Hz model: accepts an encode() output, a dict or a list of
        dicts with key "audio_codes" (torch or numpy, (T, Q) each or a padded (B, T, Q) tensor)."""
        def _to_tensor(x, dtype=None):
            if isinstance(x, torch.Tensor):
                return x
            t = torch.from_numpy(np.asarray(x))
            return t.to(dtype) if dtype is not None else t

        if hasattr(encoded, "audio_codes"):
            audio_codes_list = encoded.audio_codes
        elif isinstance(encoded, dict):
            audio_codes_list = encoded["audio_codes"]
        elif isinstance(encoded, list):
            audio_codes_list = [e["audio_codes"] for e in encoded]
        else:
            raise TypeError("`encoded` must be an encode output, a dict, or a list of dicts.")   # IT:313
        if isinstance(audio_codes_list, torch.Tensor):
            t = audio_codes_list
            if t.dim() == 1:
                t = t.unsqueeze(0)
            elif t.dim() == 2:
                t = t.unsqueeze(0)
            padded = t.to(self.device)
        else:
            lst = [_to_tensor(c, dtype=torch.long) for c in audio_codes_list]
            padded = pad_sequence(lst, batch_first=True, padding_value=-1).to(self.device)
        dec = self.model.decode(padded, return_dict=True)
        wavs = [w.to(torch.float32).detach().cpu().numpy() for w in dec.audio_values]
        return wavs, int(self.model.get_output_sample_rate())

    def get_model_type(self) -> str:
        return self.model.get_model_type()

    def get_input_sample_rate(self) -> int:
        return int(self.model.get_input_sample_rate())

    def get_output_sample_rate(self) -> int:
        return int(self.model.get_output_sample_rate())

    def get_encode_downsample_rate(self) -> int:
        return int(self.model.get_encode_downsample_rate())

    def get_decode_upsample_rate(self) -> int:
        return int(self.model.get_decode_upsample_rate())
