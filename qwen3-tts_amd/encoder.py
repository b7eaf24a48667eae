"""Codec ENCODER engine (waveform -> Qwen3-TTS-Tokenizer-12Hz codes), SURVEY.md 8(f3).

Host mirror of `Qwen3TTSTokenizerV2Model.encode` (tokenizer v2:961-991) over `qtts_encoder_*` (include/qtts.h).
Validated on MI355X in round 2: codes bit-identical to the reference's own encoder class on the golden waveforms
(tests/test_gpu_parity.py::test_codec_encoder_codes_vs_reference_golden).
"""
import ctypes as C
import threading
from typing import Any, Dict, List

import torch

from . import _lib
from .config import CodecEncoderConfig


def _relative_encoder_state(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Full tokenizer checkpoints hold the Mimi model under `encoder.` (tokenizer v2:941); the engine binds names
    relative to it ("encoder.layers.0.conv.weight", "encoder_transformer...", "downsample...", "quantizer...")."""
    roots = ("encoder.layers.", "encoder_transformer.", "downsample.", "quantizer.semantic_", "quantizer.acoustic_")
    if any(k.startswith("encoder.encoder.") for k in sd):
        sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    return {k: v for k, v in sd.items() if k.startswith(roots)}


class CodecEncoderEngine:
    """Owns one `qtts_encoder` handle."""

    def __init__(self, config: Any, state_dict: Dict[str, torch.Tensor], compute_dtype: torch.dtype = torch.float32,
                 device: str = "cuda:0", max_batch: int = 4, max_samples: int = 24000 * 30):
        self.config = CodecEncoderConfig.from_any(config)
        self.device = _lib.hip_device(device, "CodecEncoderEngine")
        self.max_batch, self.max_samples = int(max_batch), int(max_samples)
        self._lib = _lib.load_library()
        self._lock = threading.RLock()
        c = self.config
        ec = _lib.EncoderConfigC()
        for f in ("hidden_size", "num_filters", "num_residual_layers", "kernel_size", "last_kernel_size", "residual_kernel_size",
                  "dilation_growth_rate", "compress", "codebook_size", "codebook_dim", "num_quantizers",
                  "num_semantic_quantizers", "num_hidden_layers", "intermediate_size", "num_attention_heads",
                  "num_key_value_heads", "head_dim", "sliding_window"):
            setattr(ec, f, int(getattr(c, f)))
        ec.n_ratios = len(c.upsampling_ratios)
        for i, r in enumerate(c.upsampling_ratios):
            ec.ratios[i] = int(r)
        ec.valid_num_quantizers = int(c.encoder_valid_num_quantizers)
        ec.rope_theta, ec.norm_eps = float(c.rope_theta), float(c.norm_eps)
        ec.compute_dtype = _lib.QTTS_BF16 if compute_dtype == torch.bfloat16 else _lib.QTTS_F32
        ec.max_batch, ec.max_samples = self.max_batch, self.max_samples
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.qtts_encoder_create(C.byref(ec), C.byref(self._h)))
            for name, t in _relative_encoder_state(state_dict).items():
                if name.endswith("codebook.initialized"):
                    continue
                _lib.bind_tensor(self._lib.qtts_encoder_bind, self._h, name, t)
            _lib.check(self._lib.qtts_encoder_finalize(self._h))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.qtts_encoder_destroy(h)
            self._h = None

    def frames(self, samples: int) -> int:
        n = C.c_int64()
        _lib.check(self._lib.qtts_encoder_frames(self._h, int(samples), C.byref(n)))
        return int(n.value)

    @_lib.locked
    def encode_padded(self, input_values: torch.Tensor) -> torch.Tensor:
        """input_values (B, samples) float, zero-padded rows -> codes (B, valid_num_quantizers, frames) int64."""
        if input_values.dim() != 2:
            raise ValueError(f"input_values must be (batch, samples), got {tuple(input_values.shape)}")
        B, L = input_values.shape
        if B > self.max_batch or L > self.max_samples:
            raise ValueError(f"encode: batch {B} / samples {L} exceed max_batch {self.max_batch} / max_samples {self.max_samples}")
        x = input_values.to(self.device, torch.float32).contiguous()
        codes = torch.empty(B, self.config.encoder_valid_num_quantizers, self.frames(L), dtype=torch.long, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.qtts_encoder_encode(self._h, C.c_void_p(x.data_ptr()), B, L, C.c_void_p(codes.data_ptr()),
                                                     C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return codes

    def encode(self, input_values: torch.Tensor, padding_mask: torch.Tensor) -> List[torch.Tensor]:
        """Qwen3TTSTokenizerV2Model.encode (tokenizer v2:961-991): per row, the first ceil(valid_samples /
        encode_downsample_rate) frames, transposed to (frames, valid_num_quantizers)."""
        codes = self.encode_padded(input_values)
        lens = padding_mask.detach().to("cpu").numpy().sum(axis=1)      # (numpy: single-threaded -- see codec.py Qwen3TTSTokenizer.encode)
        out = []
        for c, n_valid in zip(codes, lens):
            n = -(-int(n_valid) // self.config.encode_downsample_rate)
            out.append(c[..., :n].transpose(0, 1))
        return out
