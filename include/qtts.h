/*
 * qtts.h -- C ABI of the MI355X-native Qwen3-TTS hot path (libqtts.so, gfx950 only).
 *
 * The reference (QwenLM/Qwen3-TTS) is 100 % Python and has no FFI of its own; its drop-in
 * boundary is the Python API of `qwen_tts.inference`.  This header is the native
 * seam *below* that API.  Each entry point replaces one reference call (cited per function);
 * the Python host code in `qwen3-tts_amd/` keeps the reference's names/arguments and calls these
 * through ctypes (INTEGRATION.md shows the binding a reference maintainer would add).
 *
 * Rules of the boundary
 *   - plain C types only: pointers, sizes, POD structs.  No torch / HIP types in signatures
 *     (`stream` is a `hipStream_t` passed as `void*`; NULL = the default stream).
 *   - return 0 (QTTS_OK) or a negative QTTS_ERR_*; qtts_last_error() gives the message of the
 *     last failure on the calling thread.  No exceptions cross the ABI.
 *   - weights are bound from HOST memory in the reference's own state_dict layout and naming
 *     (SURVEY.md Appendix B); the library repacks them once into its streaming layouts in HBM.
 *   - activations / codes / waveforms are DEVICE pointers owned by the caller (PyTorch-ROCm
 *     allocations in the shipped host code); the library borrows them for the call only.
 *   - one handle per (process, device); a handle is not re-entrant (the reference keeps
 *     per-call state on the module too: modeling_qwen3_tts.py:1704 `self.rope_deltas`).
 */
#ifndef QTTS_H
#define QTTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QTTS_OK 0
#define QTTS_ERR_ARG (-1)     /* bad argument / shape                                      */
#define QTTS_ERR_HIP (-2)     /* a HIP runtime call failed                                 */
#define QTTS_ERR_STATE (-3)   /* call order violated (e.g. decode before finalize/prefill) */
#define QTTS_ERR_UNBOUND (-4) /* finalize(): a required weight was never bound             */
#define QTTS_ERR_NAME (-5)    /* bind(): unknown parameter name                            */
#define QTTS_ERR_LIMIT (-6)   /* exceeds max_batch / max_seq given at create               */

/* storage / arithmetic type of weights and KV cache inside the engine */
#define QTTS_F32 0  /* parity mode: fp32 weights, exact-f32 MFMA (v_mfma_f32_16x16x4_f32)   */
#define QTTS_BF16 1 /* perf mode:   bf16 weights + KV, v_mfma_f32_16x16x32_bf16, f32 accum  */

const char* qtts_last_error(void);
/* ABI version of this header; bumped on any signature change (2: + qtts_talker_text_embed, qtts_talker_assemble_rows;
 * 3: + qtts_codec_stream_begin, qtts_codec_stream_push; 4: + qtts_encoder_*; 5: + qtts_speaker_*;
 * 6: + qtts_talker_stream_*; 7: + qtts_talker_set_teacher; 8: + qtts_talker_set_profile / get_gemm_profile;
 * 9: + qtts_codec_get_stats; 10: + qtts_set_option / qtts_get_option, qtts_talker_stats grew the fused-launch fields;
 * 11: + qtts_talker_debug_cp_logits, qtts_talker_stats.cp_layer_per_step in the reserved word;
 * 12: + qtts_talker_stats.ks_split_per_step (appended)). */
#define QTTS_ABI_VERSION 12
int qtts_abi_version(void);

/* A/B switches of the library (measuring tools and tests; a deployment sets none).  Every switch has a name of the form
 * "QTTS_..." (the list: DESIGN.md "switches"); its value is, in this order, what qtts_set_option last gave it, else the
 * environment variable of the same name as it was when the library first looked, else the built-in default.  Engine-level
 * switches are copied into a handle when it is CREATED (qtts_*_create): later changes do not touch existing handles.
 * Launcher-level switches (kernel selection inside csrc/) are looked up per launch and follow the table at once.
 * value = NULL removes the override.  No reference counterpart (the reference has no native code to switch). */
int qtts_set_option(const char* name, const char* value);
/* -> the current value ("" + return 1 when the switch is unset), copied into buf (cap bytes incl. the terminator). */
int qtts_get_option(const char* name, char* buf, int32_t cap);

/* ------------------------------------------------------------------------------------------
 * Codec decoder: Qwen3-TTS-Tokenizer-12Hz  codes -> 24 kHz waveform.
 * Replaces Qwen3TTSTokenizerV2Model.decode / Qwen3TTSTokenizerV2Decoder.{forward,chunked_decode}
 * (qwen_tts/core/tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py:993-1024, 869-896).
 * ------------------------------------------------------------------------------------------ */
typedef struct qtts_codec qtts_codec;

typedef struct {
    /* fields of Qwen3TTSTokenizerV2DecoderConfig (configuration_qwen3_tts_tokenizer_v2.py:72-93) */
    int32_t codebook_size;
    int32_t codebook_dim;
    int32_t hidden_size;
    int32_t latent_dim;
    int32_t num_attention_heads;
    int32_t num_key_value_heads;
    int32_t head_dim;
    int32_t sliding_window;
    int32_t intermediate_size;
    int32_t num_hidden_layers;
    int32_t num_quantizers;
    int32_t n_upsample_rates;
    int32_t upsample_rates[8];
    int32_t n_upsampling_ratios;
    int32_t upsampling_ratios[8];
    int32_t decoder_dim;
    float rms_norm_eps;
    float rope_theta;
    /* engine options */
    int32_t compute_dtype; /* QTTS_F32 | QTTS_BF16 */
    int32_t max_batch;     /* largest B of one decode call                     */
    int32_t max_frames;    /* largest frames per *chunk* (reference: 300 + 25) */
} qtts_codec_config;

int qtts_codec_create(const qtts_codec_config* cfg, qtts_codec** out);
void qtts_codec_destroy(qtts_codec* c);

/* Bind one parameter.  `name` is the reference state_dict key relative to `decoder.`
 * (e.g. "quantizer.rvq_first.vq.layers.0._codebook.embedding_sum", "decoder.1.block.1.conv.weight").
 * `host` points to HOST memory, row-major, `src_dtype` QTTS_F32 or QTTS_BF16.  The data is
 * copied/repacked before the call returns. */
int qtts_codec_bind(qtts_codec* c, const char* name, const void* host, int32_t src_dtype,
                    int32_t ndim, const int64_t* shape);
/* Verify every parameter is bound, precompute derived tables (normalised codebooks V2:677,
 * exp(alpha), 1/(exp(beta)+1e-9) V2:608-613) and allocate workspaces. */
int qtts_codec_finalize(qtts_codec* c);

/* One un-chunked Qwen3TTSTokenizerV2Decoder.forward (V2:869-884).
 * codes_dev: int64 (B, Q, T) device, values in [0, codebook_size).
 * wav_dev:   float (B, T*total_upsample) device.
 * pre_clamp_dev: optional float (B, T*total_upsample) device, the tensor before clamp(-1,1). */
int qtts_codec_forward(qtts_codec* c, const int64_t* codes_dev, int32_t B, int32_t T, float* wav_dev,
                       float* pre_clamp_dev, void* stream);

/* Qwen3TTSTokenizerV2Model.decode (V2:993-1024): codes_dev int64 (B, T, Q) device padded with -1,
 * clamp(min=0), chunked_decode(chunk_size, left_context) (V2:886-896; reference defaults 300, 25),
 * wav_dev float (B, T*total_upsample) device.  lengths_host (B) receives
 * (#frames with code > -1) * total_upsample -- the caller trims, as V2:1017 does. */
int qtts_codec_decode(qtts_codec* c, const int64_t* codes_dev, int32_t B, int32_t T, int32_t chunk_size,
                      int32_t left_context, float* wav_dev, int64_t* lengths_host, void* stream);

/* Streaming (state-carrying) decode -- SURVEY.md 8(f2).  The decoder is causal end to end
 * (Qwen3TTSTokenizerV2CausalConvNet / CausalTransConvNet, tokenizer v2:159-208; sliding-window transformer :491), so
 * packets of new frames can be decoded one after the other while the handle carries, per stateful layer, the last
 * (k-1)*dilation input rows (one row per k = 2r transposed conv, window-1 rotated q|k|v rows per transformer layer).
 * The concatenated packets equal Qwen3TTSTokenizerV2Decoder.forward (:869-884) on the whole sequence -- without
 * chunked_decode's (:886-896) re-decode of 25 context frames; algorithm in oracle/codec_stream_ref.py.
 * One session per handle: `begin` zeroes the state for `batch` sequences that advance in lockstep; `push` decodes
 * codes_dev int64 (batch, num_quantizers, n_frames) into wav_dev float (batch, n_frames * total_upsample).
 * STATUS: validated on MI355X (round 2): any packetisation == `forward` on the whole sequence
 * (tests/test_gpu_parity.py::test_codec_incremental_stream_equals_forward). */
/* Bookkeeping of the decode-call graph cache (round 4): `qtts_codec_forward` / `qtts_codec_decode` replay a captured hipGraph from
 * the second call with the same shape (B, T, chunking) on, staging codes and waveform in engine-owned buffers -- the reference's decode is a Python loop of
 * eager PyTorch ops (tokenizer v2:869-896); there is nothing to mirror, this only reports what happened. */
typedef struct qtts_codec_stats {
    int32_t graph_captures;  /* decode shapes captured so far                                   */
    int32_t graph_replays;   /* calls served by hipGraphLaunch                                  */
    int32_t graphs_cached;   /* captured graphs alive (LRU of 8)                                */
    int32_t graph_nodes_last; /* nodes of the most recently REPLAYED graph (0: nothing replayed yet)      */
} qtts_codec_stats;
int qtts_codec_get_stats(qtts_codec* c, qtts_codec_stats* out);

int qtts_codec_stream_begin(qtts_codec* c, int32_t batch);
int qtts_codec_stream_push(qtts_codec* c, const int64_t* codes_dev, int32_t n_frames, float* wav_dev, void* stream);

/* Test/diagnostic hook: run Qwen3TTSTokenizerV2Decoder.forward on codes_dev (B, Q, T) up to and including
 * `stage` and copy that stage's activation, channel-last float (B, L, C), to out_dev (capacity `cap` floats).
 * Stage names: "rvq","pre_conv","pre_transformer","upsample0","upsample1","decoder0","block1".."block4". */
int qtts_codec_forward_stage(qtts_codec* c, const int64_t* codes_dev, int32_t B, int32_t T, const char* stage,
                             float* out_dev, int64_t cap, int64_t* L, int64_t* C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Codec ENCODER: 24 kHz waveform -> Qwen3-TTS-Tokenizer-12Hz codes -- SURVEY.md 8(f3).
 * Replaces Qwen3TTSTokenizerV2Model.encode (tokenizer v2:961-991), i.e. transformers.MimiModel.encode behind
 * Qwen3TTSTokenizerV2Encoder (v2:897-908): SEANet encoder -> causal transformer -> stride-2 downsample -> split
 * residual VQ; only the first `valid_num_quantizers` codebooks are produced (v2:982-983).
 * STATUS: validated on MI355X (round 2): codes bit-identical to the reference's own encoder class on the golden waveforms.
 * ------------------------------------------------------------------------------------------ */
typedef struct qtts_encoder qtts_encoder;

typedef struct {
    /* encoder-side fields of transformers.MimiConfig as the reference instantiates it */
    int32_t hidden_size;
    int32_t num_filters;
    int32_t num_residual_layers;
    int32_t n_ratios;
    int32_t ratios[8];              /* MimiConfig.upsampling_ratios (the encoder applies them reversed) */
    int32_t kernel_size;
    int32_t last_kernel_size;
    int32_t residual_kernel_size;
    int32_t dilation_growth_rate;
    int32_t compress;
    int32_t codebook_size;
    int32_t codebook_dim;
    int32_t num_quantizers;
    int32_t num_semantic_quantizers;
    int32_t valid_num_quantizers;   /* Qwen3TTSTokenizerV2Config.encoder_valid_num_quantizers (16) */
    int32_t num_hidden_layers;
    int32_t intermediate_size;
    int32_t num_attention_heads;
    int32_t num_key_value_heads;
    int32_t head_dim;
    int32_t sliding_window;
    float rope_theta;
    float norm_eps;
    /* engine options */
    int32_t compute_dtype;          /* QTTS_F32 | QTTS_BF16 (the quantiser always runs in fp32) */
    int32_t max_batch;
    int32_t max_samples;            /* longest waveform of one encode call */
} qtts_encoder_config;

int qtts_encoder_create(const qtts_encoder_config* cfg, qtts_encoder** out);
void qtts_encoder_destroy(qtts_encoder* e);
/* `name` = reference state_dict key relative to `encoder.` (e.g. "encoder.layers.0.conv.weight",
 * "quantizer.acoustic_residual_vector_quantizer.layers.3.codebook.embed_sum").  Host pointer, row-major. */
int qtts_encoder_bind(qtts_encoder* e, const char* name, const void* host, int32_t src_dtype, int32_t ndim,
                      const int64_t* shape);
int qtts_encoder_finalize(qtts_encoder* e);
/* frames produced for a waveform of `samples` samples (MimiModel.get_encoded_length) */
int qtts_encoder_frames(qtts_encoder* e, int64_t samples, int64_t* frames);
/* wav_dev float (B, samples) device, zero-padded rows; codes_dev int64 (B, valid_num_quantizers, frames) device.
 * The caller trims each row to ceil(valid_samples / encode_downsample_rate) frames and transposes, as v2:984-985. */
int qtts_encoder_encode(qtts_encoder* e, const float* wav_dev, int32_t B, int32_t samples, int64_t* codes_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Speaker embedding of the Base (voice-clone) model -- SURVEY.md 8(f4).
 * Replaces Qwen3TTSForConditionalGeneration.extract_speaker_embedding (modeling_qwen3_tts.py:1941-1954):
 * mel_spectrogram (:402-464; n_fft 1024, hop 256, 128 mels, 0..12 kHz, center=False) -> Qwen3TTSSpeakerEncoder
 * (ECAPA-TDNN, :95-393).
 * STATUS: validated on MI355X (round 2) against the oracle (log-mel and embedding); the Slaney filterbank restates
 * librosa.filters.mel, which is absent here: filterbank parity unpinned.
 * ------------------------------------------------------------------------------------------ */
typedef struct qtts_speaker qtts_speaker;

typedef struct {
    /* Qwen3TTSSpeakerEncoderConfig (configuration_qwen3_tts.py:22-67) */
    int32_t mel_dim;
    int32_t enc_dim;
    int32_t n_blocks;               /* len(enc_channels) */
    int32_t channels[8];
    int32_t kernel_sizes[8];
    int32_t dilations[8];
    int32_t attention_channels;
    int32_t res2net_scale;
    int32_t se_channels;
    /* mel front end (the constants of modeling_qwen3_tts.py:1944-1952) */
    int32_t n_fft;
    int32_t hop_size;
    int32_t win_size;
    int32_t num_mels;
    /* engine options */
    int32_t compute_dtype;          /* QTTS_F32 | QTTS_BF16 (the STFT and mel projection always run in fp32) */
    int32_t max_batch;
    int32_t max_samples;
} qtts_speaker_config;

int qtts_speaker_create(const qtts_speaker_config* cfg, qtts_speaker** out);
void qtts_speaker_destroy(qtts_speaker* s);
/* `name` = reference state_dict key relative to `speaker_encoder.` (e.g. "blocks.1.res2net_block.blocks.0.conv.weight"),
 * plus "mel_basis" float (num_mels, n_fft/2+1): the Slaney filterbank the reference takes from librosa.filters.mel. */
int qtts_speaker_bind(qtts_speaker* s, const char* name, const void* host, int32_t src_dtype, int32_t ndim,
                      const int64_t* shape);
int qtts_speaker_finalize(qtts_speaker* s);
int qtts_speaker_mel_frames(qtts_speaker* s, int64_t samples, int64_t* frames);
/* wav_dev float (B, samples) device, 24 kHz, in [-1, 1]; emb_dev float (B, enc_dim); mels_dev optional float
 * (B, mel_frames, mel_dim) (the log-mel features, for tests). */
int qtts_speaker_embed(qtts_speaker* s, const float* wav_dev, int32_t B, int32_t samples, float* emb_dev, float* mels_dev,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Talker + code predictor: the autoregressive speech-token decoder.
 * Replaces `self.talker.generate(inputs_embeds, attention_mask, trailing_text_hidden,
 * tts_pad_embed, **talker_kwargs)` (modeling_qwen3_tts.py:2272-2278) = HF GenerationMixin._sample
 * around Qwen3TTSTalkerForConditionalGeneration.forward (modeling_qwen3_tts.py:1636-1744) and its
 * nested code_predictor.generate (modeling_qwen3_tts.py:1671-1680, 1250-1312).
 * ------------------------------------------------------------------------------------------ */
typedef struct qtts_talker qtts_talker;

typedef struct {
    /* Qwen3TTSTalkerConfig (configuration_qwen3_tts.py:370-454) */
    int32_t vocab_size;
    int32_t hidden_size;
    int32_t intermediate_size;
    int32_t num_hidden_layers;
    int32_t num_attention_heads;
    int32_t num_key_value_heads;
    int32_t head_dim;
    float rms_norm_eps;
    float rope_theta;
    int32_t num_code_groups;
    int32_t text_hidden_size;
    int32_t codec_eos_token_id;
    /* Qwen3TTSTalkerCodePredictorConfig (configuration_qwen3_tts.py:189-258) */
    int32_t cp_vocab_size;
    int32_t cp_hidden_size;
    int32_t cp_intermediate_size;
    int32_t cp_num_hidden_layers;
    int32_t cp_num_attention_heads;
    int32_t cp_num_key_value_heads;
    int32_t cp_head_dim;
    float cp_rms_norm_eps;
    float cp_rope_theta;
    /* engine options */
    int32_t weight_dtype; /* QTTS_F32 | QTTS_BF16 (weights and KV cache)            */
    int32_t max_batch;    /* sequences per generate call                              */
    int32_t max_seq;      /* prompt + generated frames per sequence (KV pages reserved) */
    int32_t use_graph;    /* 1: replay the frame step as a hipGraph; 0: eager launches  */
} qtts_talker_config;

/* HF sampling knobs (generation defaults: qwen_tts/inference/qwen3_tts_model.py:319-352). */
typedef struct {
    int32_t do_sample;
    int32_t top_k;   /* 0 = off */
    float top_p;     /* 1.0 = off */
    float temperature;
    float repetition_penalty;
    int32_t subtalker_dosample;
    int32_t subtalker_top_k;
    float subtalker_top_p;
    float subtalker_temperature;
    uint64_t seed;   /* Philox key; torch's global-RNG stream cannot be reproduced across devices */
} qtts_sampling;

int qtts_talker_create(const qtts_talker_config* cfg, qtts_talker** out);
void qtts_talker_destroy(qtts_talker* t);
/* `name` = reference state_dict key relative to `talker.` (e.g. "model.layers.0.self_attn.q_proj.weight",
 * "code_predictor.lm_head.3.weight").  Host pointer, row-major. */
int qtts_talker_bind(qtts_talker* t, const char* name, const void* host, int32_t src_dtype, int32_t ndim,
                     const int64_t* shape);
int qtts_talker_finalize(qtts_talker* t);

/* y = text_projection(x): Qwen3TTSTalkerResizeMLP (modeling_qwen3_tts.py:808-816,1575-1577), used by
 * the prompt assembly (modeling_qwen3_tts.py:2076-2232).  x_dev float (rows, text_hidden) device,
 * y_dev float (rows, hidden) device. */
int qtts_talker_text_projection(qtts_talker* t, const float* x_dev, int32_t rows, float* y_dev, void* stream);

/* Prompt assembly on device (Qwen3TTSForConditionalGeneration.generate, modeling_qwen3_tts.py:2076-2269, and
 * generate_icl_prompt, :1968-2019).  The host resolves WHICH rows make up each prompt (integers only); the device
 * does every gather, projection and sum.
 *
 * qtts_talker_text_embed: y[r] = text_projection(text_embedding[ids[r]]) (:2076-2080, 2177, 2207, 2229) --
 *   ids_dev int64 (rows) device, y_dev float (rows, hidden) device.  Needs "model.text_embedding.weight" and the
 *   text_projection weights bound.  An id outside the table is QTTS_ERR_ARG.
 * qtts_talker_assemble_rows: out[r] = T + C with, from desc_dev int32 (rows, 4) = {text_row, codec_id, spk_row,
 *   ref_frame} (-1 = absent; all four absent = a zero row, i.e. left padding :2251-2254):
 *     T = proj[text_row]                               (proj_dev float (proj_rows, hidden): text_embed output)
 *     C = codec_embedding[codec_id]                    (:2142-2172 codec prefix / pad / bos, speaker ids :2092)
 *       | spk[spk_row]                                 (spk_dev float (n_spk, hidden): voice-clone x-vectors :1957-1966)
 *       | codec_embedding[ref[f][0]] + sum_g cp_embedding[g-1][ref[f][g]]   (ref_codes_dev int64 (n_ref_frames, G):
 *                                                        in-context reference codes :1983-1990)
 *   out_dev float (rows, hidden).  Indices out of range are QTTS_ERR_ARG. */
int qtts_talker_text_embed(qtts_talker* t, const int64_t* ids_dev, int32_t rows, float* y_dev, void* stream);
int qtts_talker_assemble_rows(qtts_talker* t, const int32_t* desc_dev, int32_t rows, const float* proj_dev, int32_t proj_rows,
                              const float* spk_dev, int32_t n_spk, const int64_t* ref_codes_dev, int32_t n_ref_frames,
                              float* out_dev, void* stream);

/* Prefill (modeling_qwen3_tts.py:1665-1667, 1693-1727): embeds_dev float (B, T, H) LEFT-padded,
 * n_pad_host (B) = number of left pads per row (attention_mask = [0]*n_pad + [1]*(T-n_pad),
 * modeling_qwen3_tts.py:2251-2254), trailing_dev float (B, Tt, H) right-padded with tts_pad
 * (modeling_qwen3_tts.py:2255-2269), tts_pad_dev float (H).  Fills the KV cache and leaves the
 * first-step logits / past_hidden on device. */
int qtts_talker_prefill(qtts_talker* t, const float* embeds_dev, int32_t B, int32_t T, const int32_t* n_pad_host,
                        const float* trailing_dev, int32_t Tt, const float* tts_pad_dev, void* stream);

/* The sampling loop after prefill, with HF `_sample` semantics:
 *   processors RepetitionPenalty -> MinNewTokensLength(min_new_tokens, eos) -> SuppressTokens ->
 *   [Temperature -> TopK -> TopP], finished rows keep receiving eos, stop when every row hit eos or
 *   max_new_tokens tokens were drawn.
 * suppress_host: list of suppressed token ids (modeling_qwen3_tts.py:2059-2063), n_suppress entries.
 * codes_dev:  int64 (B, max_new_tokens-1, num_code_groups) device, row-major; frame f of row b holds
 *             [cb0, 15 sub-codes] (modeling_qwen3_tts.py:1681); frames >= *n_frames_host are undefined.
 * hidden_dev: optional float (B, max_new_tokens-1, H): `past_hidden` per frame (modeling_qwen3_tts.py:2281).
 * tokens_dev: optional int64 (B, max_new_tokens): every sampled cb0 token incl. the final one.
 * n_frames_host: number of complete frames (= HF steps - 1, modeling_qwen3_tts.py:2280).
 * Synchronises the stream before returning. */
int qtts_talker_generate(qtts_talker* t, const qtts_sampling* sp, int32_t max_new_tokens, int32_t min_new_tokens,
                         int32_t eos_token_id, const int32_t* suppress_host, int32_t n_suppress,
                         int64_t* codes_dev, float* hidden_dev, int64_t* tokens_dev, int32_t* n_frames_host,
                         void* stream);

/* Resumable generation -- qtts_talker_generate in three calls, for streaming OUTPUT (BASELINE config 4; the reference
 * only "simulates streaming text input", qwen3_tts_model.py:513-515, and returns whole utterances):
 *   stream_begin  after qtts_talker_prefill: same arguments as qtts_talker_generate minus the outputs that only exist at
 *                 the end; samples the first token.  codes_dev int64 (B, max_new_tokens-1, G) / hidden_dev (optional)
 *                 are filled progressively.
 *   stream_step   runs up to max_frames_now more frame steps (the same captured frame graph and device-resident loop
 *                 state as qtts_talker_generate) and returns, on the host, how many frames are final so far
 *                 (codes_dev[:, :frames_total]) and whether the stop condition has latched.
 *   stream_end    the closing bookkeeping: tokens_dev int64 (B, max_new_tokens) padded with -1 (optional) and the frame
 *                 count, as qtts_talker_generate reports them.  May be called early to abandon a request.
 * STATUS: validated on MI355X (round 2), eager and hipGraph: packets == one-shot generate. */
int qtts_talker_stream_begin(qtts_talker* t, const qtts_sampling* sp, int32_t max_new_tokens, int32_t min_new_tokens,
                             int32_t eos_token_id, const int32_t* suppress_host, int32_t n_suppress, int64_t* codes_dev,
                             float* hidden_dev, void* stream);
int qtts_talker_stream_step(qtts_talker* t, int32_t max_frames_now, int32_t* frames_total_host, int32_t* finished_host,
                            void* stream);
int qtts_talker_stream_end(qtts_talker* t, int64_t* tokens_dev, int32_t* n_frames_host, void* stream);

/* Test/diagnostic hooks (device -> caller device buffers, after prefill / a generate call). */
int qtts_talker_debug_logits(qtts_talker* t, float* logits_dev /* (B, vocab) */, void* stream);
/* The code predictor's RAW logits of the last frame step that ran, every pass: what `code_predictor.generate`'s lm_head[j] returned
 * before HF's processors (modeling_qwen3_tts.py:1250-1312; the sampled path is checked against the processed softmax of exactly
 * these numbers, tests/test_gpu_parity.py).  The frame step keeps one row block per pass, so nothing is added to it.  ABI v11. */
int qtts_talker_debug_cp_logits(qtts_talker* t, float* logits_dev /* (num_code_groups - 1, B, cp_vocab) */, void* stream);

/* Per-call statistics of the last generate (for bench.py): kernel-side byte model inputs. */
typedef struct {
    int32_t frames_run;       /* frame steps executed on device (>= n_frames, poll granularity)   */
    int32_t graph_nodes;      /* kernel nodes in the captured frame step (0 if eager)              */
    double weight_bytes_per_frame; /* bytes of packed weights one frame step streams              */
    double gemm_ms_last;      /* HIP-event time of the dominant kernel class over the call, if enabled */
    int64_t gemm_launches_last;
    int32_t long_graphs;      /* long-sequence frame graphs currently cached (one per KV-length bucket the generation reached, ABI v8) */
    int32_t attn_nsplit_last; /* split-KV workgroups per (sequence, kv head) of the last launched frame step (1 = short mode) */
    int32_t attn_span_last;   /* the key span those workgroups partition (the live length's bucket; 0 = short mode) */
    /* the code predictor's fused launch (q|k|v + attention + o-projection of a layer in one launch; ABI v10) */
    int32_t cp_fused_per_step;      /* fused attention + o-projection launches in the frame step last launched / captured (0: separate launches; bf16 only) */
    int64_t cp_fused_launches_last; /* = cp_fused_per_step x frames_run of the last generation                                              */
    int32_t cp_fused_giveups;       /* generations that ended with QTTS_ERR_STATE because a consumer gave up (the engine then left the fused launch) */
    int32_t cp_fused_capacity;      /* engines with the code predictor's fused launches the DEVICE holds at once (register-share account, talker_engine.hip) */
    int32_t cp_fused_active;        /* 1: this engine holds one of those places                                                             */
    int32_t cp_mlp_per_step;        /* fused MLP launches (cp_mlp.hip: gate|up + SwiGLU + down of a code-predictor layer; bf16 and fp32) in that frame step */
    int32_t cp_layer_per_step;      /* of those, the launches that ran BOTH stages of a layer as one (cp_layer.hip, round 6): they count in cp_fused_per_step
                                     * and cp_mlp_per_step too.  (ABI v11: the former reserved word.)                                        */
    int32_t ks_split_per_step;      /* decode GEMMs of that frame step that split K over workgroups and combine inside the launch (skinny.hip: skinny2_ks_kernel;
                                     * bf16 engines at batch 17..32: the o- / down-projections; ABI v12)                                       */
    int32_t reserved0;
} qtts_talker_stats;
int qtts_talker_get_stats(qtts_talker* t, qtts_talker_stats* out);
/* Per-class result of the profile mode (qtts_talker_set_profile(t, 1), ABI v8): every launch of the decode GEMM in frames 1..6
 * of the last generate call, timed on its own with the kernel's begin / end timestamps, grouped by (stack, N, K). */
typedef struct {
    int32_t stack;            /* 0 talker layers, 1 code predictor (layers, projection, lm_head), 2 talker codec_head */
    int32_t N, K;             /* GEMM shape: out[M <= 64][N] = x[M][K] . W[N][K]^T */
    int64_t launches;
    double total_ms;          /* sum of the launches' kernel durations */
    double min_us, max_us;
    double bytes_per_launch;  /* algorithmic bytes: the packed weight matrix, N * K * element size */
} qtts_gemm_class;
int qtts_talker_get_gemm_profile(qtts_talker* t, qtts_gemm_class* out, int32_t cap, int32_t* n);
/* Teacher forcing (diagnostic mode for parity measurements; no reference counterpart -- it is how a whole utterance of the
 * bf16 mode is compared decision by decision with the reference's greedy run, SURVEY.md 7 "teacher-forced per-step logits"):
 * until disabled (forced_codes_dev = NULL) every following qtts_talker_generate call -- greedy, min_new_tokens ==
 * max_new_tokens == n_frames + 1, run eagerly -- records the engine's OWN greedy choice of every codebook of every frame into
 * own_dev (B, n_frames + 1, G) int32 ([b][i][0] = own cb-0 token i, [b][f][1 + j] = own sub-code j of frame f) and then
 * continues with forced_codes_dev (B, n_frames, G) int64 instead (frame-level forcing: the 15-pass code predictor runs free
 * inside a frame; its result is replaced before the embedding sum, M:1681-1692; the fed cb-0 token and the repetition-penalty
 * history follow the forced sequence).  logit_slots_dev (n_frames + 1) int32 maps a token step to a slot of
 * logits_trace_dev (n_slots, B, vocab) fp32 that receives the raw cb-0 logits of that step (-1 = not traced); both may be
 * NULL.  All pointers are device pointers owned by the caller and must stay valid while the mode is on. */
int qtts_talker_set_teacher(qtts_talker* t, const int64_t* forced_codes_dev, int32_t n_frames, int32_t* own_dev,
                            const int32_t* logit_slots_dev, float* logits_trace_dev);
/* bench.py's roofline leg.  enable = 1: the next qtts_talker_generate runs its frame steps eagerly and times EVERY launch of the
 * dominant kernel (skinny weight-streaming decode GEMM) of frames 1..6 of the REAL frame step on its own (kernel begin / end
 * timestamps through hipExtLaunchKernelGGL events); results per GEMM class from qtts_talker_get_gemm_profile.  enable = 2:
 * round 2's measurement (the GEMM launches of one frame step replayed in isolation as a hipGraph; the call produces no
 * usable codes).  0 = off. */
int qtts_talker_set_profile(qtts_talker* t, int32_t enable);

#ifdef __cplusplus
}
#endif
#endif /* QTTS_H */
