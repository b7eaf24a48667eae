"""Plain-dataclass views of the reference's configuration objects.

The reference keeps its dimensions in HF `PretrainedConfig` subclasses
(qwen_tts/core/models/configuration_qwen3_tts.py:189-258,370-502 and
qwen_tts/core/tokenizer_12hz/configuration_qwen3_tts_tokenizer_v2.py:72-172).  The engine only *reads*
them ("every kernel must take dims from the live config"), so here they are plain
dataclasses that can be built from a `config.json` dict, from an HF config object, or by hand.
Defaults are the reference's class defaults.
"""
from dataclasses import dataclass, field, fields
from typing import Any, Dict, Optional, Tuple


def _get(src: Any, key: str, default=None):
    if isinstance(src, dict):
        return src.get(key, default)
    return getattr(src, key, default)


def _pick(cls, src: Any, **renames) -> Dict[str, Any]:
    out = {}
    for f in fields(cls):
        k = renames.get(f.name, f.name)
        v = _get(src, k, None)
        if v is not None:
            out[f.name] = v
    return out


@dataclass
class CodecDecoderConfig:
    """Qwen3TTSTokenizerV2DecoderConfig (configuration_qwen3_tts_tokenizer_v2.py:72-123)."""
    codebook_size: int = 2048
    codebook_dim: int = 512               # arrives only via **kwargs in the reference (tokenizer v2:831)
    hidden_size: int = 1024
    latent_dim: int = 1024
    max_position_embeddings: int = 8000
    rope_theta: float = 10000.0
    num_attention_heads: int = 16
    num_key_value_heads: int = 16
    head_dim: Optional[int] = None        # defaults to hidden_size // num_attention_heads (tokenizer v2:289)
    sliding_window: int = 72
    intermediate_size: int = 3072
    layer_scale_initial_scale: float = 0.01
    rms_norm_eps: float = 1e-5
    num_hidden_layers: int = 8
    num_quantizers: int = 16
    upsample_rates: Tuple[int, ...] = (8, 5, 4, 3)
    upsampling_ratios: Tuple[int, ...] = (2, 2)
    decoder_dim: int = 1536
    # Qwen3TTSTokenizerV2Config level (configuration_qwen3_tts_tokenizer_v2.py:143-170)
    output_sample_rate: int = 24000
    input_sample_rate: int = 24000
    decode_upsample_rate: int = 1920
    encode_downsample_rate: int = 1920

    def __post_init__(self):
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_attention_heads
        self.upsample_rates = tuple(int(x) for x in self.upsample_rates)
        self.upsampling_ratios = tuple(int(x) for x in self.upsampling_ratios)

    @property
    def total_upsample(self) -> int:
        n = 1
        for r in self.upsample_rates + self.upsampling_ratios:
            n *= r
        return n

    @classmethod
    def from_any(cls, src: Any) -> "CodecDecoderConfig":
        """`src`: this class, a dict / HF object of the decoder config, or of the full tokenizer config
        (with `decoder_config` inside)."""
        if isinstance(src, cls):
            return src
        dec = _get(src, "decoder_config", None)
        kw = _pick(cls, dec if dec is not None else src)
        for k in ("output_sample_rate", "input_sample_rate", "decode_upsample_rate", "encode_downsample_rate"):
            v = _get(src, k, None)
            if v is not None:
                kw[k] = v
        return cls(**kw)


@dataclass
class CodecEncoderConfig:
    """Encoder-side view of the tokenizer config: `encoder_config` is a transformers.MimiConfig dict
    (configuration_qwen3_tts_tokenizer_v2.py:143-170), plus the two call-site fields of tokenizer v2:940-944."""
    hidden_size: int = 512
    num_filters: int = 64
    num_residual_layers: int = 1
    upsampling_ratios: Tuple[int, ...] = (8, 6, 5, 4)
    kernel_size: int = 7
    last_kernel_size: int = 3
    residual_kernel_size: int = 3
    dilation_growth_rate: int = 2
    compress: int = 2
    codebook_size: int = 2048
    codebook_dim: int = 256
    num_quantizers: int = 32
    num_semantic_quantizers: int = 1
    num_hidden_layers: int = 8
    intermediate_size: int = 2048
    num_attention_heads: int = 8
    num_key_value_heads: int = 8
    head_dim: Optional[int] = None
    sliding_window: int = 250
    rope_theta: float = 10000.0
    norm_eps: float = 1e-5
    encoder_valid_num_quantizers: int = 16
    encode_downsample_rate: int = 1920
    input_sample_rate: int = 24000

    def __post_init__(self):
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_attention_heads
        self.upsampling_ratios = tuple(int(x) for x in self.upsampling_ratios)

    @classmethod
    def from_any(cls, src: Any) -> "CodecEncoderConfig":
        """`src`: this class, a dict / object of the full tokenizer config (with `encoder_config`), or of the Mimi
        config itself (dataclass-style flat dicts work too)."""
        if isinstance(src, cls):
            return src
        enc = _get(src, "encoder_config", None)
        kw = _pick(cls, enc if enc is not None else src)
        rp = _get(enc if enc is not None else src, "rope_parameters", None)       # transformers 5.x spelling
        if rp is not None and _get(rp, "rope_theta", None) is not None:
            kw["rope_theta"] = float(_get(rp, "rope_theta"))
        for k in ("encoder_valid_num_quantizers", "encode_downsample_rate", "input_sample_rate"):
            v = _get(src, k, None)
            if v is not None:
                kw[k] = v
        return cls(**kw)


@dataclass
class TalkerConfig:
    """Qwen3TTSTalkerConfig + its code_predictor_config + the token ids of Qwen3TTSConfig."""
    vocab_size: int = 3072
    hidden_size: int = 1024
    intermediate_size: int = 2048
    num_hidden_layers: int = 20
    num_attention_heads: int = 16
    num_key_value_heads: int = 2
    head_dim: Optional[int] = None        # only via **kwargs in the reference (M:734)
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    num_code_groups: int = 32
    text_hidden_size: int = 2048
    text_vocab_size: int = 151936         # only via **kwargs in the reference (M:1442)
    codec_eos_token_id: int = 4198
    codec_think_id: int = 4202
    codec_nothink_id: int = 4203
    codec_think_bos_id: int = 4204
    codec_think_eos_id: int = 4205
    codec_pad_id: int = 4196
    codec_bos_id: int = 4197
    spk_id: Dict[str, int] = field(default_factory=dict)
    spk_is_dialect: Dict[str, Any] = field(default_factory=dict)
    codec_language_id: Dict[str, int] = field(default_factory=dict)
    # code predictor (Qwen3TTSTalkerCodePredictorConfig defaults)
    cp_vocab_size: int = 2048
    cp_hidden_size: int = 1024
    cp_intermediate_size: int = 3072
    cp_num_hidden_layers: int = 5
    cp_num_attention_heads: int = 16
    cp_num_key_value_heads: int = 8
    cp_head_dim: int = 128
    cp_rms_norm_eps: float = 1e-6
    cp_rope_theta: float = 10000.0
    # Qwen3TTSConfig level
    im_start_token_id: int = 151644
    im_end_token_id: int = 151645
    tts_pad_token_id: int = 151671
    tts_bos_token_id: int = 151672
    tts_eos_token_id: int = 151673
    tts_model_type: Optional[str] = None
    tts_model_size: Optional[str] = None
    tokenizer_type: Optional[str] = None

    def __post_init__(self):
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_attention_heads
        self.spk_id = dict(self.spk_id or {})
        self.spk_is_dialect = dict(self.spk_is_dialect or {})
        self.codec_language_id = dict(self.codec_language_id or {})

    @classmethod
    def from_any(cls, src: Any) -> "TalkerConfig":
        """`src`: this class, or a dict / HF object of the top-level Qwen3TTSConfig (with `talker_config`)
        or of the talker config itself."""
        if isinstance(src, cls):
            return src
        tk = _get(src, "talker_config", None)
        top = src if tk is not None else None
        tk = tk if tk is not None else src
        kw = {}
        for f in fields(cls):      # flat `cp_*` keys are accepted too (dataclass dicts); the nested config wins
            v = _get(tk, f.name, None)
            if v is not None:
                kw[f.name] = v
        cp = _get(tk, "code_predictor_config", None)
        if cp is not None:
            for f in fields(cls):
                if f.name.startswith("cp_"):
                    v = _get(cp, f.name[3:], None)
                    if v is not None:
                        kw[f.name] = v
        if top is not None:
            for k in ("im_start_token_id", "im_end_token_id", "tts_pad_token_id", "tts_bos_token_id",
                      "tts_eos_token_id", "tts_model_type", "tts_model_size", "tokenizer_type"):
                v = _get(top, k, None)
                if v is not None:
                    kw[k] = v
        return cls(**kw)
