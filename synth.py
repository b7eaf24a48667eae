"""Synthetic workload generator (configurations, seeded weights, prompts) -- shared by bench.py, tests/, tools/ and the
golden generation under oracle/.  It contains NO model arithmetic and nothing of the reference's algorithm: it only
fills reference-shaped tensors with seeded random numbers (it is neither the product nor the oracle).

Deterministic synthetic configurations and weights for the Qwen3-TTS hot path.

No checkpoint exists offline (SURVEY.md headline fact 3), so every parity and
performance run uses seeded random weights at real (or tiny) shapes.  A weight is a
pure function of (seed, parameter name, shape): numpy PCG64 seeded with
[seed, crc32(name)] -- identical in the build container (golden generation against
the reference) and on the GPU box (parity tests against the committed goldens),
so weights never have to travel.

Parameter names / shapes follow the reference modules' state_dict (SURVEY.md
Appendix B; modeling_qwen3_tts.py:1427-1445,1571-1583,1019-1032,1163-1174 and
tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py:824-865).
"""
import zlib
from dataclasses import dataclass, field, asdict
from typing import Dict, Tuple

import numpy as np


# --------------------------------------------------------------------------- configs
@dataclass
class CodecCfg:
    """Dims of Qwen3TTSTokenizerV2DecoderConfig (configuration_qwen3_tts_tokenizer_v2.py:72-93)."""
    codebook_size: int = 2048
    codebook_dim: int = 512          # VQ dim = codebook_dim // 2 (V2:831)
    hidden_size: int = 1024
    latent_dim: int = 1024
    num_attention_heads: int = 16
    num_key_value_heads: int = 16
    head_dim: int = 64
    sliding_window: int = 72
    intermediate_size: int = 3072
    num_hidden_layers: int = 8
    num_quantizers: int = 16
    upsample_rates: Tuple[int, ...] = (8, 5, 4, 3)
    upsampling_ratios: Tuple[int, ...] = (2, 2)
    decoder_dim: int = 1536
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_position_embeddings: int = 8000

    @property
    def total_upsample(self) -> int:
        return int(np.prod(self.upsample_rates + self.upsampling_ratios))


@dataclass
class TalkerCfg:
    """Dims of Qwen3TTSTalkerConfig + its code_predictor_config (configuration_qwen3_tts.py:189-258,370-454)."""
    vocab_size: int = 3072
    hidden_size: int = 1024
    intermediate_size: int = 3072
    num_hidden_layers: int = 28
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    num_code_groups: int = 16
    text_hidden_size: int = 2048
    text_vocab_size: int = 151936
    # code predictor
    cp_vocab_size: int = 2048
    cp_hidden_size: int = 1024
    cp_intermediate_size: int = 3072
    cp_num_hidden_layers: int = 5
    cp_num_attention_heads: int = 16
    cp_num_key_value_heads: int = 8
    cp_head_dim: int = 128
    cp_rms_norm_eps: float = 1e-6
    cp_rope_theta: float = 1000000.0
    # special codec ids (all inside [vocab-1024, vocab); synthetic values)
    codec_eos_token_id: int = 2150
    codec_think_id: int = 2154
    codec_nothink_id: int = 2155
    codec_think_bos_id: int = 2156
    codec_think_eos_id: int = 2157
    codec_pad_id: int = 2148
    codec_bos_id: int = 2149
    # text special ids (Qwen3TTSConfig, configuration_qwen3_tts.py:466-470)
    im_start_token_id: int = 151644
    im_end_token_id: int = 151645
    tts_pad_token_id: int = 151671
    tts_bos_token_id: int = 151672
    tts_eos_token_id: int = 151673
    spk_id: Dict[str, int] = field(default_factory=lambda: {"vivian": 3000, "ryan": 3001})
    spk_is_dialect: Dict[str, object] = field(default_factory=lambda: {"vivian": False, "ryan": False})
    codec_language_id: Dict[str, int] = field(default_factory=lambda: {"chinese": 2050, "english": 2051})


def codec_real() -> CodecCfg:
    return CodecCfg()


def codec_tiny() -> CodecCfg:
    # head_dim kept at the real 64; channel counts multiples of 32 all the way down
    return CodecCfg(codebook_size=64, codebook_dim=64, hidden_size=128, latent_dim=64,
                    num_attention_heads=2, num_key_value_heads=2, head_dim=64, sliding_window=8,
                    intermediate_size=160, num_hidden_layers=2, decoder_dim=512)


def talker_06b() -> TalkerCfg:
    return TalkerCfg()


def talker_17b() -> TalkerCfg:
    return TalkerCfg(hidden_size=2048, intermediate_size=6144)


def talker_tiny() -> TalkerCfg:
    # talker hidden != code-predictor hidden so small_to_mtp_projection is exercised (M:1171-1174)
    return TalkerCfg(vocab_size=1280, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                     num_attention_heads=4, num_key_value_heads=2, head_dim=128, rope_theta=10000.0,
                     cp_rope_theta=10000.0,
                     text_hidden_size=192, text_vocab_size=512,
                     cp_vocab_size=256, cp_hidden_size=128, cp_intermediate_size=256,
                     cp_num_hidden_layers=2, cp_num_attention_heads=4, cp_num_key_value_heads=2,
                     cp_head_dim=128,
                     codec_eos_token_id=358, codec_think_id=362, codec_nothink_id=363,
                     codec_think_bos_id=364, codec_think_eos_id=365, codec_pad_id=356, codec_bos_id=357,
                     im_start_token_id=500, im_end_token_id=501, tts_pad_token_id=502,
                     tts_bos_token_id=503, tts_eos_token_id=504,
                     spk_id={"vivian": 1200, "ryan": 1201},
                     spk_is_dialect={"vivian": False, "ryan": False},
                     codec_language_id={"chinese": 300, "english": 301})


# --------------------------------------------------------------------------- weights
FINAL_CONV_GAIN = 0.12


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def _normal(seed, name, shape, std=1.0, mean=0.0):
    a = _rng(seed, name).standard_normal(shape, dtype=np.float32)
    if std != 1.0:
        a *= np.float32(std)
    if mean != 0.0:
        a += np.float32(mean)
    return a


def _uniform(seed, name, shape, lo, hi):
    return _rng(seed, name).uniform(lo, hi, shape).astype(np.float32)


def codec_param_shapes(c: CodecCfg) -> Dict[str, tuple]:
    """Name -> shape of every `decoder.*`-relative parameter the decode path reads."""
    s = {}
    vq = c.codebook_dim // 2
    for q, n in (("rvq_first", 1), ("rvq_rest", c.num_quantizers - 1)):
        for i in range(n):
            s[f"quantizer.{q}.vq.layers.{i}._codebook.embedding_sum"] = (c.codebook_size, vq)
            s[f"quantizer.{q}.vq.layers.{i}._codebook.cluster_usage"] = (c.codebook_size,)
        s[f"quantizer.{q}.output_proj.weight"] = (c.codebook_dim, vq, 1)
        s[f"quantizer.{q}.input_proj.weight"] = (vq, c.codebook_dim, 1)   # present, unused by decode
    s["pre_conv.conv.weight"] = (c.latent_dim, c.codebook_dim, 3)
    s["pre_conv.conv.bias"] = (c.latent_dim,)
    H, I = c.hidden_size, c.intermediate_size
    qd = c.num_attention_heads * c.head_dim
    kvd = c.num_key_value_heads * c.head_dim
    s["pre_transformer.input_proj.weight"] = (H, c.latent_dim)
    s["pre_transformer.input_proj.bias"] = (H,)
    s["pre_transformer.output_proj.weight"] = (c.latent_dim, H)
    s["pre_transformer.output_proj.bias"] = (c.latent_dim,)
    s["pre_transformer.norm.weight"] = (H,)
    for l in range(c.num_hidden_layers):
        p = f"pre_transformer.layers.{l}."
        s[p + "self_attn.q_proj.weight"] = (qd, H)
        s[p + "self_attn.k_proj.weight"] = (kvd, H)
        s[p + "self_attn.v_proj.weight"] = (kvd, H)
        s[p + "self_attn.o_proj.weight"] = (H, qd)
        s[p + "mlp.gate_proj.weight"] = (I, H)
        s[p + "mlp.up_proj.weight"] = (I, H)
        s[p + "mlp.down_proj.weight"] = (H, I)
        s[p + "input_layernorm.weight"] = (H,)
        s[p + "post_attention_layernorm.weight"] = (H,)
        s[p + "self_attn_layer_scale.scale"] = (H,)
        s[p + "mlp_layer_scale.scale"] = (H,)
    L = c.latent_dim
    for u, f in enumerate(c.upsampling_ratios):
        s[f"upsample.{u}.0.conv.weight"] = (L, L, f)          # ConvTranspose1d: (in, out, k)
        s[f"upsample.{u}.0.conv.bias"] = (L,)
        s[f"upsample.{u}.1.dwconv.conv.weight"] = (L, 1, 7)
        s[f"upsample.{u}.1.dwconv.conv.bias"] = (L,)
        s[f"upsample.{u}.1.norm.weight"] = (L,)
        s[f"upsample.{u}.1.norm.bias"] = (L,)
        s[f"upsample.{u}.1.pwconv1.weight"] = (4 * L, L)
        s[f"upsample.{u}.1.pwconv1.bias"] = (4 * L,)
        s[f"upsample.{u}.1.pwconv2.weight"] = (L, 4 * L)
        s[f"upsample.{u}.1.pwconv2.bias"] = (L,)
        s[f"upsample.{u}.1.gamma"] = (L,)
    D = c.decoder_dim
    s["decoder.0.conv.weight"] = (D, L, 7)
    s["decoder.0.conv.bias"] = (D,)
    for i, r in enumerate(c.upsample_rates):
        cin, cout = D // 2 ** i, D // 2 ** (i + 1)
        p = f"decoder.{i + 1}.block."
        s[p + "0.alpha"] = (cin,)
        s[p + "0.beta"] = (cin,)
        s[p + "1.conv.weight"] = (cin, cout, 2 * r)            # ConvTranspose1d: (in, out, k)
        s[p + "1.conv.bias"] = (cout,)
        for j in (2, 3, 4):
            for a in ("act1", "act2"):
                s[p + f"{j}.{a}.alpha"] = (cout,)
                s[p + f"{j}.{a}.beta"] = (cout,)
            s[p + f"{j}.conv1.conv.weight"] = (cout, cout, 7)
            s[p + f"{j}.conv1.conv.bias"] = (cout,)
            s[p + f"{j}.conv2.conv.weight"] = (cout, cout, 1)
            s[p + f"{j}.conv2.conv.bias"] = (cout,)
    n = len(c.upsample_rates)
    cl = D // 2 ** n
    s[f"decoder.{n + 1}.alpha"] = (cl,)
    s[f"decoder.{n + 1}.beta"] = (cl,)
    s[f"decoder.{n + 2}.conv.weight"] = (1, cl, 7)
    s[f"decoder.{n + 2}.conv.bias"] = (1,)
    return s


def _codec_value(seed, name, shape, c: CodecCfg):
    """Distribution per parameter kind.  Conv/Linear weights use std = 1/sqrt(fan_in) so that
    activations stay O(1) through the 40-odd layers (plain N(0,0.02) saturates the final clamp,
    SURVEY.md 8d 'fixture caveat'); LayerScale/gamma are drawn large enough that an error inside
    the transformer / ConvNeXt branch is visible in the waveform."""
    n = len(c.upsample_rates)
    if name.endswith("embedding_sum"):
        return _normal(seed, name, shape)
    if name.endswith("cluster_usage"):
        return _uniform(seed, name, shape, 0.5, 2.0)
    if name.endswith(".alpha") or name.endswith(".beta"):
        return _normal(seed, name, shape, 0.3)
    if name.endswith("gamma"):
        return _normal(seed, name, shape, 0.1, 0.3)
    if name.endswith("layer_scale.scale"):
        return _normal(seed, name, shape, 0.05, 0.25)
    if "layernorm.weight" in name or name.endswith("norm.weight"):
        return _normal(seed, name, shape, 0.02, 1.0)
    if name.endswith(".bias"):
        return _normal(seed, name, shape, 0.02)
    if name.endswith(".weight"):
        if ".block.1.conv." in name or (name.startswith("upsample.") and ".0.conv." in name):
            fan_in = shape[0] * 2 if ".block.1." in name else shape[0]   # transposed conv: 2 taps (or 1) hit each output
        else:
            fan_in = int(np.prod(shape[1:]))
        std = 1.0 / np.sqrt(fan_in)
        if ".conv2.conv.weight" in name:
            std *= 0.35                                            # residual branch gain: keep the 12 res units from compounding
        if name == f"decoder.{n + 2}.conv.weight":
            std *= FINAL_CONV_GAIN                                 # pre-clamp rms ~0.4: most samples inside (-1,1), a few clamped
        return _normal(seed, name, shape, std)
    raise KeyError(name)


def codec_weights(c: CodecCfg, seed: int = 1234) -> Dict[str, np.ndarray]:
    return {k: _codec_value(seed, k, shp, c) for k, shp in codec_param_shapes(c).items()}


def talker_param_shapes(t: TalkerCfg, with_text: bool = True) -> Dict[str, tuple]:
    """Name -> shape of every `talker.*`-relative parameter on the path."""
    s = {}
    H, I = t.hidden_size, t.intermediate_size
    qd, kvd = t.num_attention_heads * t.head_dim, t.num_key_value_heads * t.head_dim
    for l in range(t.num_hidden_layers):
        p = f"model.layers.{l}."
        s[p + "self_attn.q_proj.weight"] = (qd, H)
        s[p + "self_attn.k_proj.weight"] = (kvd, H)
        s[p + "self_attn.v_proj.weight"] = (kvd, H)
        s[p + "self_attn.o_proj.weight"] = (H, qd)
        s[p + "self_attn.q_norm.weight"] = (t.head_dim,)
        s[p + "self_attn.k_norm.weight"] = (t.head_dim,)
        s[p + "mlp.gate_proj.weight"] = (I, H)
        s[p + "mlp.up_proj.weight"] = (I, H)
        s[p + "mlp.down_proj.weight"] = (H, I)
        s[p + "input_layernorm.weight"] = (H,)
        s[p + "post_attention_layernorm.weight"] = (H,)
    s["model.norm.weight"] = (H,)
    s["model.codec_embedding.weight"] = (t.vocab_size, H)
    s["codec_head.weight"] = (t.vocab_size, H)
    if with_text:
        s["model.text_embedding.weight"] = (t.text_vocab_size, t.text_hidden_size)
        s["text_projection.linear_fc1.weight"] = (t.text_hidden_size, t.text_hidden_size)
        s["text_projection.linear_fc1.bias"] = (t.text_hidden_size,)
        s["text_projection.linear_fc2.weight"] = (H, t.text_hidden_size)
        s["text_projection.linear_fc2.bias"] = (H,)
    ch, ci = t.cp_hidden_size, t.cp_intermediate_size
    cq, ckv = t.cp_num_attention_heads * t.cp_head_dim, t.cp_num_key_value_heads * t.cp_head_dim
    for l in range(t.cp_num_hidden_layers):
        p = f"code_predictor.model.layers.{l}."
        s[p + "self_attn.q_proj.weight"] = (cq, ch)
        s[p + "self_attn.k_proj.weight"] = (ckv, ch)
        s[p + "self_attn.v_proj.weight"] = (ckv, ch)
        s[p + "self_attn.o_proj.weight"] = (ch, cq)
        s[p + "self_attn.q_norm.weight"] = (t.cp_head_dim,)
        s[p + "self_attn.k_norm.weight"] = (t.cp_head_dim,)
        s[p + "mlp.gate_proj.weight"] = (ci, ch)
        s[p + "mlp.up_proj.weight"] = (ci, ch)
        s[p + "mlp.down_proj.weight"] = (ch, ci)
        s[p + "input_layernorm.weight"] = (ch,)
        s[p + "post_attention_layernorm.weight"] = (ch,)
    s["code_predictor.model.norm.weight"] = (ch,)
    for g in range(t.num_code_groups - 1):
        s[f"code_predictor.model.codec_embedding.{g}.weight"] = (t.cp_vocab_size, H)   # talker width (M:1030-1032)
        s[f"code_predictor.lm_head.{g}.weight"] = (t.cp_vocab_size, ch)
    if ch != H:
        s["code_predictor.small_to_mtp_projection.weight"] = (ch, H)
        s["code_predictor.small_to_mtp_projection.bias"] = (ch,)
    return s


def _talker_value(seed, name, shape):
    """N(0, 0.02) Linear/Embedding as in the reference init (M:512-523) but with non-zero biases and
    perturbed norm weights so a dropped bias / norm weight cannot hide."""
    if name.endswith("norm.weight") or "layernorm.weight" in name:
        return _normal(seed, name, shape, 0.02, 1.0)
    if name.endswith(".bias"):
        return _normal(seed, name, shape, 0.02)
    if name.startswith("codec_head.") or ".lm_head." in name:
        return _normal(seed, name, shape, 0.08)     # 4x wider logits: greedy margins well above fp32 noise
    return _normal(seed, name, shape, 0.02)


def talker_weights(t: TalkerCfg, seed: int = 1234, with_text: bool = True) -> Dict[str, np.ndarray]:
    return {k: _talker_value(seed, k, shp) for k, shp in talker_param_shapes(t, with_text).items()}


# ----------------------------------------------------------------------------- speaker encoder
@dataclass
class SpeakerCfg:
    """Qwen3TTSSpeakerEncoderConfig (configuration_qwen3_tts.py:22-67)."""
    mel_dim: int = 128
    enc_dim: int = 1024
    enc_channels: tuple = (512, 512, 512, 512, 1536)
    enc_kernel_sizes: tuple = (5, 3, 3, 3, 1)
    enc_dilations: tuple = (1, 2, 3, 4, 1)
    enc_attention_channels: int = 128
    enc_res2net_scale: int = 8
    enc_se_channels: int = 128
    sample_rate: int = 24000


def speaker_real(enc_dim: int = 2048) -> SpeakerCfg:
    return SpeakerCfg(enc_dim=enc_dim)          # enc_dim = talker hidden size (the x-vector is a prompt row, M:2092)


def speaker_tiny() -> SpeakerCfg:
    return SpeakerCfg(mel_dim=16, enc_dim=24, enc_channels=(32, 32, 32, 32, 96), enc_attention_channels=8,
                      enc_res2net_scale=4, enc_se_channels=8)


def speaker_small() -> SpeakerCfg:
    """Smallest configuration the HIP speaker encoder accepts (every contraction a multiple of 32, real mel front end)."""
    return SpeakerCfg(mel_dim=128, enc_dim=64, enc_channels=(128, 128, 128, 128, 384), enc_attention_channels=32,
                      enc_res2net_scale=4, enc_se_channels=32)


def speaker_param_shapes(c: SpeakerCfg) -> Dict[str, tuple]:
    """state_dict names / shapes of Qwen3TTSSpeakerEncoder (modeling_qwen3_tts.py:312-367), relative to `speaker_encoder.`."""
    ch, ks = list(c.enc_channels), list(c.enc_kernel_sizes)
    out: Dict[str, tuple] = {}

    def conv(name, cout, cin, k):
        out[name + ".weight"] = (cout, cin, k)
        out[name + ".bias"] = (cout,)

    conv("blocks.0.conv", ch[0], c.mel_dim, ks[0])
    for i in range(1, len(ch) - 1):
        p = f"blocks.{i}."
        conv(p + "tdnn1.conv", ch[i], ch[i - 1], 1)
        for j in range(c.enc_res2net_scale - 1):
            conv(p + f"res2net_block.blocks.{j}.conv", ch[i] // c.enc_res2net_scale, ch[i] // c.enc_res2net_scale, ks[i])
        conv(p + "tdnn2.conv", ch[i], ch[i], 1)
        conv(p + "se_block.conv1", c.enc_se_channels, ch[i], 1)
        conv(p + "se_block.conv2", ch[i], c.enc_se_channels, 1)
    conv("mfa.conv", ch[-1], ch[-1], ks[-1])
    conv("asp.tdnn.conv", c.enc_attention_channels, ch[-1] * 3, 1)
    conv("asp.conv", ch[-1], c.enc_attention_channels, 1)
    conv("fc", c.enc_dim, ch[-1] * 2, 1)
    return out


def speaker_weights(c: SpeakerCfg, seed: int = 1234) -> Dict[str, np.ndarray]:
    out = {}
    for k, shp in speaker_param_shapes(c).items():
        if k.endswith(".bias"):
            out[k] = _normal(seed, k, shp, 0.05)
        else:
            fan_in = shp[1] * shp[2]
            out[k] = _normal(seed, k, shp, 1.0 / np.sqrt(fan_in))
    return out


# ----------------------------------------------------------------------------- codec encoder (Mimi)
@dataclass
class MimiEncCfg:
    """The encoder-side fields of transformers.MimiConfig as the reference uses it (tokenizer v2:897-908) plus the two
    Qwen3TTSTokenizerV2Config fields of the call site (v2:940-944)."""
    sampling_rate: int = 24000
    audio_channels: int = 1
    hidden_size: int = 512
    num_filters: int = 64
    num_residual_layers: int = 1
    upsampling_ratios: tuple = (8, 6, 5, 4)
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
    head_dim: int = 64
    sliding_window: int = 250
    rope_theta: float = 10000.0
    norm_eps: float = 1e-5
    encoder_valid_num_quantizers: int = 16
    encode_downsample_rate: int = 1920


def mimi_enc_real() -> MimiEncCfg:
    return MimiEncCfg()


def mimi_enc_tiny() -> MimiEncCfg:
    return MimiEncCfg(hidden_size=32, num_filters=4, upsampling_ratios=(4, 2), codebook_size=16, codebook_dim=8,
                      num_quantizers=6, num_hidden_layers=2, intermediate_size=48, num_attention_heads=2,
                      num_key_value_heads=2, head_dim=16, sliding_window=6, encoder_valid_num_quantizers=4,
                      encode_downsample_rate=16)


def mimi_enc_small() -> MimiEncCfg:
    """Smallest configuration whose every contraction is a multiple of 32 and whose head_dim the attention kernel
    supports: what the HIP encoder's parity test runs."""
    return MimiEncCfg(hidden_size=128, num_filters=64, upsampling_ratios=(4, 2), codebook_size=64, codebook_dim=32,
                      num_quantizers=6, num_hidden_layers=2, intermediate_size=256, num_attention_heads=2,
                      num_key_value_heads=2, head_dim=64, sliding_window=6, encoder_valid_num_quantizers=4,
                      encode_downsample_rate=16)


def mimi_enc_param_shapes(c: MimiEncCfg) -> Dict[str, tuple]:
    """Encoder-side state_dict of MimiModel (modeling_mimi.py MimiEncoder / MimiTransformerModel / downsample /
    MimiSplitResidualVectorQuantizer), names relative to `encoder.` of the Qwen3-TTS tokenizer checkpoint."""
    out: Dict[str, tuple] = {}

    def conv(name, cout, cin, k, bias=True):
        out[name + ".conv.weight"] = (cout, cin, k)
        if bias:
            out[name + ".conv.bias"] = (cout,)

    idx, ch = 0, c.num_filters
    conv(f"encoder.layers.{idx}", ch, c.audio_channels, c.kernel_size)
    idx += 1
    for ratio in reversed(c.upsampling_ratios):
        for _ in range(c.num_residual_layers):
            conv(f"encoder.layers.{idx}.block.1", ch // c.compress, ch, c.residual_kernel_size)
            conv(f"encoder.layers.{idx}.block.3", ch, ch // c.compress, 1)
            idx += 1
        idx += 1                                                   # ELU
        conv(f"encoder.layers.{idx}", ch * 2, ch, 2 * ratio)
        idx += 1
        ch *= 2
    idx += 1                                                       # ELU
    conv(f"encoder.layers.{idx}", c.hidden_size, ch, c.last_kernel_size)
    H, I, qd, kvd = c.hidden_size, c.intermediate_size, c.num_attention_heads * c.head_dim, c.num_key_value_heads * c.head_dim
    for l in range(c.num_hidden_layers):
        p = f"encoder_transformer.layers.{l}."
        out[p + "self_attn.q_proj.weight"] = (qd, H)
        out[p + "self_attn.k_proj.weight"] = (kvd, H)
        out[p + "self_attn.v_proj.weight"] = (kvd, H)
        out[p + "self_attn.o_proj.weight"] = (H, qd)
        out[p + "mlp.fc1.weight"] = (I, H)
        out[p + "mlp.fc2.weight"] = (H, I)
        for n in ("input_layernorm", "post_attention_layernorm"):
            out[p + n + ".weight"] = (H,)
            out[p + n + ".bias"] = (H,)
        out[p + "self_attn_layer_scale.scale"] = (H,)
        out[p + "mlp_layer_scale.scale"] = (H,)
    out["downsample.conv.weight"] = (H, H, 4)
    for name, n in (("semantic", c.num_semantic_quantizers), ("acoustic", c.num_quantizers - c.num_semantic_quantizers)):
        p = f"quantizer.{name}_residual_vector_quantizer."
        out[p + "input_proj.weight"] = (c.codebook_dim, H, 1)
        out[p + "output_proj.weight"] = (H, c.codebook_dim, 1)
        for i in range(n):
            out[p + f"layers.{i}.codebook.initialized"] = (1,)
            out[p + f"layers.{i}.codebook.cluster_usage"] = (c.codebook_size,)
            out[p + f"layers.{i}.codebook.embed_sum"] = (c.codebook_size, c.codebook_dim)
    return out


def mimi_enc_weights(c: MimiEncCfg, seed: int = 1234) -> Dict[str, np.ndarray]:
    out = {}
    for k, shp in mimi_enc_param_shapes(c).items():
        if k.endswith("initialized"):
            out[k] = np.ones(shp, np.float32)
        elif k.endswith("cluster_usage"):
            out[k] = _uniform(seed, k, shp, 0.5, 2.0)
        elif k.endswith("embed_sum"):
            out[k] = _normal(seed, k, shp, 1.0)
        elif k.endswith("layer_scale.scale"):
            out[k] = _normal(seed, k, shp, 0.05, 0.3)
        elif k.endswith("layernorm.weight"):
            out[k] = _normal(seed, k, shp, 0.05, 1.0)
        elif k.endswith(".bias"):
            out[k] = _normal(seed, k, shp, 0.05)
        else:
            fan_in = int(np.prod(shp[1:]))
            out[k] = _normal(seed, k, shp, 1.3 / np.sqrt(fan_in))
    return out


def weights_checksum(w: Dict[str, np.ndarray]) -> float:
    """Cheap order-independent fingerprint stored in goldens to prove both sides built the same weights."""
    acc = 0.0
    for k in sorted(w):
        a = w[k].ravel()
        acc += float(np.float64(a[:: max(1, a.size // 4096)].astype(np.float64).sum())) * (1 + (zlib.crc32(k.encode()) % 7))
    return acc


def cfg_dict(c) -> dict:
    return asdict(c)


def rand_audio(seed: int, batch: int, samples: int, sr: int = 24000) -> np.ndarray:
    """Seeded speech-shaped test audio (batch, samples) float32 in (-1, 1): a few harmonics of a drifting f0 under a slow amplitude
    envelope plus a little noise.  Shared by oracle/gen_golden.py (reference run) and the tests, so the waveform itself is never stored."""
    g = np.random.default_rng(seed)
    t = np.arange(samples, dtype=np.float64) / sr
    out = np.zeros((batch, samples), np.float64)
    for b in range(batch):
        f0 = g.uniform(100.0, 280.0) * (1.0 + 0.08 * np.sin(2 * np.pi * g.uniform(1.0, 3.0) * t + g.uniform(0, 6.28)))
        ph = 2 * np.pi * np.cumsum(f0) / sr
        x = sum(g.uniform(0.2, 1.0) / (h + 1) * np.sin((h + 1) * ph + g.uniform(0, 6.28)) for h in range(6))
        env = 0.55 + 0.45 * np.sin(2 * np.pi * g.uniform(2.0, 5.0) * t + g.uniform(0, 6.28))
        x = 0.25 * x * env + 0.01 * g.standard_normal(samples)
        out[b] = x
    return np.clip(out, -0.99, 0.99).astype(np.float32)


def rand_prompt(g: np.random.Generator, t: TalkerCfg, lens, n_trail: int, scale: float = 0.05):
    """Synthetic inputs at the `talker.generate` seam: ragged LEFT-padded embeds (B,T,H), mask (B,T),
    trailing text (B,n_trail,H), tts_pad (1,1,H) as torch tensors.  Shared by gen_golden / tests / bench."""
    import torch
    B, Tm, H = len(lens), max(lens), t.hidden_size
    emb = np.zeros((B, Tm, H), np.float32)
    mask = np.zeros((B, Tm), np.int64)
    for i, l in enumerate(lens):
        emb[i, Tm - l:] = g.standard_normal((l, H), dtype=np.float32) * scale
        mask[i, Tm - l:] = 1
    trailing = g.standard_normal((B, n_trail, H), dtype=np.float32) * scale
    pad = g.standard_normal((1, 1, H), dtype=np.float32) * scale
    return torch.from_numpy(emb), torch.from_numpy(mask), torch.from_numpy(trailing), torch.from_numpy(pad)


def icl_requests(t: TalkerCfg, seed: int, text_lens, ref_text_lens, ref_frames, languages=None):
    """A batch of voice-clone (Base model, ICL) requests at the `Qwen3TTSForConditionalGeneration.generate` seam, as the wrapper
    builds them (inference/qwen3_tts_model.py:470-631): assistant-wrapped text ids, `<|im_start|>assistant\n ref text <|im_end|>\n`
    ref ids, `ref_code` (frames x 16: codebook 0 below the talker's special range, the rest below the code predictor's vocab) and
    an x-vector per request.  Shared by oracle/gen_golden.py and the GPU test so that both sides see identical requests."""
    import torch
    g = np.random.default_rng(seed)
    a, n = 77, 198
    B = len(text_lens)
    hi = min(t.text_vocab_size, t.im_start_token_id) - 1
    ids, ref_ids, ref_code, spk = [], [], [], []
    for i in range(B):
        body = g.integers(0, hi, (text_lens[i],)).tolist()
        ids.append(torch.tensor([[t.im_start_token_id, a, n] + body + [t.im_end_token_id, n, t.im_start_token_id, a, n]]))
        rbody = g.integers(0, hi, (ref_text_lens[i],)).tolist()
        ref_ids.append(torch.tensor([[t.im_start_token_id, a, n] + rbody + [t.im_end_token_id, n]]))
        nf = ref_frames[i]
        rc = np.concatenate([g.integers(0, t.vocab_size - 1024, (nf, 1)), g.integers(0, t.cp_vocab_size, (nf, t.num_code_groups - 1))], 1)
        ref_code.append(torch.from_numpy(rc))
        spk.append(torch.from_numpy(g.standard_normal(t.hidden_size).astype(np.float32) * 0.1))
    langs = languages or [("english", "auto", "chinese")[i % 3] for i in range(B)]
    vcp = dict(ref_code=ref_code, ref_spk_embedding=spk, x_vector_only_mode=[False] * B, icl_mode=[True] * B)
    return dict(ids=ids, ref_ids=ref_ids, vcp=vcp, languages=langs)
