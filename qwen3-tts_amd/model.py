"""Drop-in mirrors of the reference's top-level model and inference wrapper.

  * Qwen3TTSForConditionalGeneration.generate   qwen_tts/core/models/modeling_qwen3_tts.py:2022-2292
    (prompt assembly -> talker generate -> EOS trim).  Prompt assembly = integer row plan on the host
    (`build_prompt_plan`) + two calls into the HIP library (text embed/projection, row assembly); the decode
    loop is the HIP engine.
  * Qwen3TTSModel                               qwen_tts/inference/qwen3_tts_model.py:54-877
    (`from_pretrained`, `generate_custom_voice`, `generate_voice_design`, `generate_voice_clone`,
    kwargs merging, validation, same exception types).

SURVEY.md 8f3/8f4: building a voice-clone prompt from raw audio runs the codec *encoder* (encoder.py) and the ECAPA
speaker encoder (speaker.py); both are compiled and CPU-emulated in round 1, their first hardware run is pending, so
`create_voice_clone_prompt` is experimental, while `generate_voice_clone(voice_clone_prompt=...)` with precomputed
items is validated.
"""
import json
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from .config import TalkerConfig
from .talker import TalkerEngine


@dataclass
class VoiceClonePromptItem:
    """Same fields as the reference container (qwen3_tts_model.py:41-51)."""
    ref_code: Optional[torch.Tensor]
    ref_spk_embedding: torch.Tensor
    x_vector_only_mode: bool
    icl_mode: bool
    ref_text: Optional[str] = None


def split_packet_at_eos(packet: torch.Tensor, alive: List[bool], eos: int) -> List[torch.Tensor]:
    """Streaming form of the EOS trim of M:2283-2289: `packet` (B, k, G) are the next k frames of every request; a request
    keeps its frames up to (excluding) the first one whose first codebook is `eos`, and is finished from then on.
    `alive` is updated in place."""
    out = []
    first = packet[:, :, 0].cpu()
    for i in range(packet.shape[0]):
        if not alive[i]:
            out.append(packet[i, :0])
            continue
        hit = (first[i] == eos).nonzero()
        if hit.numel():
            alive[i] = False
            out.append(packet[i, : int(hit[0])])
        else:
            out.append(packet[i])
    return out


PLAN_BOS_ROW, PLAN_EOS_ROW, PLAN_PAD_ROW = 0, 1, 2      # rows of the projected special text tokens in every plan


def build_prompt_plan(config, input_ids, languages, speakers=None, instruct_ids=None, non_streaming_mode=False,
                      ref_ids=None, voice_clone_prompt=None) -> Dict[str, Any]:
    """Host half of the prompt assembly (M:2068-2269, generate_icl_prompt M:1968-2019): pure integer bookkeeping, no
    device work.  Every prompt / trailing row becomes a descriptor {text_row, codec_id, spk_row, ref_frame} (-1 =
    absent) that `qtts_talker_assemble_rows` turns into `text_projection(text_embedding[..]) + codec-side term`.

    Returns: text_ids int64 (R,) -- ids to embed+project, rows 0..2 = tts_bos / tts_eos / tts_pad;
    desc int32 (n*Tm + n*Tt, 4) -- left-padded prompt rows of all requests, then the trailing-text rows (padded with
    tts_pad); mask int64 (n, Tm); spk_vectors / ref_codes -- the tensors spk_row / ref_frame index into."""
    c = config
    n = len(input_ids)
    if speakers is None:
        speakers = [None] * n
    ids_np = lambda t: np.asarray(t.detach().cpu().reshape(-1).numpy() if torch.is_tensor(t) else t, dtype=np.int64).reshape(-1)

    # ---- per-request metadata first (raises before any device work, in the reference's loop order)
    spk_vectors: List[torch.Tensor] = []
    metas = []
    for i in range(n):
        speaker, language = speakers[i], languages[i]
        spk = None                                  # ("codec", id) | ("vec", row) | None
        if voice_clone_prompt is None:
            if not (speaker == "" or speaker is None):
                if speaker.lower() not in c.spk_id:
                    raise NotImplementedError(f"Speaker {speaker} not implemented")
                spk = ("codec", int(c.spk_id[speaker.lower()]))
        elif voice_clone_prompt["x_vector_only_mode"][i] or voice_clone_prompt["icl_mode"][i]:      # M:1957-1966
            spk = ("vec", len(spk_vectors))
            spk_vectors.append(voice_clone_prompt["ref_spk_embedding"][i])
        assert language is not None
        if language.lower() == "auto":
            lang_id = None
        else:
            if language.lower() not in c.codec_language_id:
                raise NotImplementedError(f"Language {language} not implemented")
            lang_id = c.codec_language_id[language.lower()]
        if (language.lower() in ["chinese", "auto"] and speaker != "" and speaker is not None
                and c.spk_is_dialect[speaker.lower()] != False):  # noqa: E712  (reference compares with != False)
            lang_id = c.codec_language_id[c.spk_is_dialect[speaker.lower()]]
        icl = bool(voice_clone_prompt is not None and voice_clone_prompt.get("ref_code") is not None
                   and voice_clone_prompt["icl_mode"][i])
        metas.append((spk, lang_id, icl))

    text_ids: List[int] = [c.tts_bos_token_id, c.tts_eos_token_id, c.tts_pad_token_id]

    def seg(ids) -> List[int]:                      # register a text segment, return its projected-row indices
        a = ids_np(ids)
        r0 = len(text_ids)
        text_ids.extend(int(x) for x in a)
        return list(range(r0, r0 + len(a)))

    T = lambda r: (r, -1, -1, -1)                   # text only
    ref_codes: List[torch.Tensor] = []
    n_ref = 0
    seqs, trails = [], []
    for i in range(n):
        spk, lang_id, icl = metas[i]
        ids = ids_np(input_ids[i])
        role = seg(ids[:3])
        if lang_id is None:
            pre = [c.codec_nothink_id, c.codec_think_bos_id, c.codec_think_eos_id]
        else:
            pre = [c.codec_think_id, c.codec_think_bos_id, lang_id, c.codec_think_eos_id]
        cin = [(-1, int(x), -1, -1) for x in pre]                                # codec prefix (M:2142-2172)
        if spk is not None:
            cin.append((-1, spk[1], -1, -1) if spk[0] == "codec" else (-1, -1, spk[1], -1))
        cin += [(-1, int(c.codec_pad_id), -1, -1), (-1, int(c.codec_bos_id), -1, -1)]
        rows = []
        if instruct_ids is not None and instruct_ids[i] is not None:
            rows += [T(r) for r in seg(instruct_ids[i])]
        rows += [T(r) for r in role]
        for k in range(len(cin) - 1):                                            # tts_pad x (len-2), tts_bos  +  cin[:-1]
            rows.append((PLAN_PAD_ROW if k < len(cin) - 2 else PLAN_BOS_ROW,) + cin[k][1:])
        if icl:
            te = seg(np.concatenate([ids_np(ref_ids[i])[3:-2], ids[3:-5]])) + [PLAN_EOS_ROW]
            rc = voice_clone_prompt["ref_code"][i]
            ce = [(-1, int(c.codec_bos_id), -1, -1)] + [(-1, -1, -1, n_ref + f) for f in range(int(rc.shape[0]))]   # M:1983-1998
            ref_codes.append(rc)
            n_ref += int(rc.shape[0])
            tl, cl = len(te), len(ce)
            if non_streaming_mode:
                rows += [(r, int(c.codec_pad_id), -1, -1) for r in te]
                rows += [(PLAN_PAD_ROW,) + e[1:] for e in ce]
                trail = [PLAN_PAD_ROW]
            elif tl > cl:
                rows += [(te[k],) + ce[k][1:] for k in range(cl)]
                trail = te[cl:]
            else:
                rows += [((te[k] if k < tl else PLAN_PAD_ROW),) + ce[k][1:] for k in range(cl)]
                trail = [PLAN_PAD_ROW]
        elif non_streaming_mode:
            rest = seg(ids[3:-5]) + [PLAN_EOS_ROW]
            rows += [(r, int(c.codec_pad_id), -1, -1) for r in rest]
            rows.append((PLAN_PAD_ROW, int(c.codec_bos_id), -1, -1))
            trail = [PLAN_PAD_ROW]
        else:
            first = seg(ids[3:4])
            rows.append((first[0],) + cin[-1][1:])
            trail = seg(ids[4:-5]) + [PLAN_EOS_ROW]
        seqs.append(rows)
        trails.append(trail)

    Tm = max(len(r) for r in seqs)
    Tt = max(len(t) for t in trails)
    desc = np.full((n * Tm + n * Tt, 4), -1, dtype=np.int32)
    mask = np.zeros((n, Tm), dtype=np.int64)
    for i, r in enumerate(seqs):
        desc[i * Tm + Tm - len(r): (i + 1) * Tm] = np.asarray(r, dtype=np.int32)
        mask[i, Tm - len(r):] = 1
    for i, t in enumerate(trails):
        base = n * Tm + i * Tt
        desc[base: base + Tt, 0] = PLAN_PAD_ROW
        desc[base: base + len(t), 0] = np.asarray(t, dtype=np.int32)
    return {"n": n, "Tm": Tm, "Tt": Tt, "text_ids": np.asarray(text_ids, dtype=np.int64), "desc": desc, "mask": mask,
            "spk_vectors": spk_vectors, "ref_codes": ref_codes}


class Qwen3TTSForConditionalGeneration:
    """Talker-side model object: owns the HIP talker engine and the embedding tables the prompt needs."""

    def __init__(self, config: Any, state_dict: Dict[str, torch.Tensor], device: str = "cuda:0",
                 dtype: torch.dtype = torch.bfloat16, max_batch: int = 8, max_seq: int = 4096,
                 use_graph: bool = True):
        self.config = TalkerConfig.from_any(config)
        self.dtype = dtype
        sd = state_dict
        if any(k.startswith("talker.") for k in sd):
            sd = {k[len("talker."):]: v for k, v in sd.items() if k.startswith("talker.")}
        self.talker = TalkerEngine(self.config, sd, weight_dtype=dtype, device=device, max_batch=max_batch,
                                   max_seq=max_seq, use_graph=use_graph)
        self.device = self.talker.device            # (the HIP device the engine accepted: `_lib.hip_device` refuses anything else)
        if "model.text_embedding.weight" not in sd:
            raise KeyError("state_dict has no talker.model.text_embedding.weight (needed by the prompt assembly)")
        self.speech_tokenizer = None
        self.generate_config = None
        self.speaker_encoder = None             # built on first use from the `speaker_encoder.*` weights (Base model)
        self.speaker_encoder_sample_rate = 24000
        full = state_dict
        self._speaker_state = {k: v for k, v in full.items() if k.startswith("speaker_encoder.")} or None
        self._speaker_config = config
        self.supported_speakers = list(self.config.spk_id.keys())
        self.supported_languages = ["auto"] + [k for k in self.config.codec_language_id if "dialect" not in k]   # M:1831-1834
        self.tokenizer_type = self.config.tokenizer_type
        self.tts_model_size = self.config.tts_model_size
        self.tts_model_type = self.config.tts_model_type

    def load_speech_tokenizer(self, speech_tokenizer):
        self.speech_tokenizer = speech_tokenizer

    def load_generate_config(self, generate_config):
        self.generate_config = generate_config

    def get_supported_speakers(self):
        return self.supported_speakers

    def get_supported_languages(self):
        return self.supported_languages

    # ------------------------------------------------------------------ prompt assembly (device, M:2068-2269)
    def assemble_prompts(self, input_ids, languages, speakers=None, instruct_ids=None, non_streaming_mode=False,
                         ref_ids=None, voice_clone_prompt=None):
        """M:2068-2269.  Returns (inputs_embeds (B,T,H) left-padded, attention_mask (B,T), trailing_text_hidden
        (B,Tt,H) right-padded with tts_pad, tts_pad_embed (1,1,H)).

        The host only decides WHICH rows make up each prompt (`build_prompt_plan`, integers); every gather,
        the text projection and every sum run in the HIP library: one `qtts_talker_text_embed` call over all text
        ids of the batch and one `qtts_talker_assemble_rows` call over all prompt + trailing rows."""
        plan = build_prompt_plan(self.config, input_ids, languages, speakers, instruct_ids, non_streaming_mode, ref_ids,
                                 voice_clone_prompt)
        H = self.config.hidden_size
        n, Tm, Tt = plan["n"], plan["Tm"], plan["Tt"]
        proj = self.talker.text_embed(torch.from_numpy(plan["text_ids"]))
        spk = torch.stack([e.reshape(-1).to(self.device, torch.float32) for e in plan["spk_vectors"]]) if plan["spk_vectors"] else None
        ref = torch.cat([r.to(self.device, torch.long) for r in plan["ref_codes"]], dim=0) if plan["ref_codes"] else None
        rows = self.talker.assemble_rows(torch.from_numpy(plan["desc"]), proj, spk, ref)
        embeds = rows[: n * Tm].reshape(n, Tm, H)
        trailing = rows[n * Tm:].reshape(n, Tt, H)
        mask = torch.from_numpy(plan["mask"]).to(self.device)
        return embeds, mask, trailing, proj[PLAN_PAD_ROW].reshape(1, 1, H)

    def extract_speaker_embedding(self, audio: np.ndarray, sr: int) -> torch.Tensor:
        """M:1941-1954: 24 kHz waveform -> (enc_dim,) x-vector, log-mel + ECAPA-TDNN on the HIP speaker engine
        (validated on MI355X in round 2: tests/test_gpu_parity.py::test_speaker_embedding_vs_oracle)."""
        assert sr == 24000, "Only support 24kHz audio"
        if not self._speaker_state:
            raise NotImplementedError("this checkpoint has no `speaker_encoder.*` weights (only the Base model does)")
        n = int(np.asarray(audio).shape[-1])
        if self.speaker_encoder is None or n > self.speaker_encoder.max_samples:
            # built on first use; re-created with a larger workspace for a longer reference (the reference has no limit)
            from .speaker import SpeakerEncoderEngine
            cap = max(30 * 24000, -(-n // 240000) * 240000)
            self.speaker_encoder = None
            self.speaker_encoder = SpeakerEncoderEngine(self._speaker_config, self._speaker_state, compute_dtype=torch.float32,
                                                        device=str(self.device), max_samples=cap)
        return self.speaker_encoder.extract_speaker_embedding(audio, sr)

    SPEAKER_BATCH = 8                       # clips per speaker-encoder call (the engine's workspace is sized for it)

    def extract_speaker_embeddings(self, audios: List[np.ndarray], sr: int) -> List[torch.Tensor]:
        """`extract_speaker_embedding` (M:1941-1954) for a LIST of 24 kHz waveforms -- what `create_voice_clone_prompt` (IM:356-458)
        computes clip by clip.  Clips of EQUAL length run through the log-mel front end and the ECAPA-TDNN as one batch of up to
        SPEAKER_BATCH rows, ragged clips are bucketed by exact length (`SpeakerEncoderEngine.embed_many`); the batched rows equal the
        clip-by-clip ones (GPU test).  The result list is in the order of `audios`."""
        assert sr == 24000, "Only support 24kHz audio"
        if not self._speaker_state:
            raise NotImplementedError("this checkpoint has no `speaker_encoder.*` weights (only the Base model does)")
        arrs = [np.asarray(a, dtype=np.float32).reshape(-1) for a in audios]
        if not arrs:
            return []
        n = max(a.shape[0] for a in arrs)
        if self.speaker_encoder is None or n > self.speaker_encoder.max_samples or self.speaker_encoder.max_batch < min(self.SPEAKER_BATCH, len(arrs)):
            from .speaker import SpeakerEncoderEngine
            cap = max(30 * 24000, -(-n // 240000) * 240000)
            if self.speaker_encoder is not None:
                cap = max(cap, self.speaker_encoder.max_samples)
            self.speaker_encoder = None
            self.speaker_encoder = SpeakerEncoderEngine(self._speaker_config, self._speaker_state, compute_dtype=torch.float32,
                                                        device=str(self.device), max_batch=self.SPEAKER_BATCH, max_samples=cap)
        return self.speaker_encoder.embed_many(arrs)

    # ------------------------------------------------------------------ generate (seam S1)
    @torch.no_grad()
    def generate(self, input_ids: Optional[List[torch.Tensor]] = None, instruct_ids: Optional[List[torch.Tensor]] = None,
                 ref_ids: Optional[List[torch.Tensor]] = None, voice_clone_prompt: Optional[dict] = None,
                 languages: List[str] = None, speakers: List[str] = None, non_streaming_mode: bool = False,
                 max_new_tokens: int = 4096, do_sample: bool = True, top_k: int = 50, top_p: float = 1.0,
                 temperature: float = 0.9, subtalker_dosample: bool = True, subtalker_top_k: int = 50,
                 subtalker_top_p: float = 1.0, subtalker_temperature: float = 0.9, eos_token_id: Optional[int] = None,
                 repetition_penalty: float = 1.05, **kwargs):
        c = self.config
        embeds, mask, trailing, pad = self.assemble_prompts(input_ids, languages, speakers, instruct_ids,
                                                            non_streaming_mode, ref_ids, voice_clone_prompt)
        suppress = [i for i in range(c.vocab_size - 1024, c.vocab_size) if i != c.codec_eos_token_id]      # M:2059-2063
        codes_all, hidden_all = [], []
        mb = self.talker.max_batch
        # one base seed per call (torch's advancing generator unless given); wave w samples with base + w so that equal rows of
        # different waves do not repeat each other's random draws (the Philox counter is (step, row-in-wave, codebook))
        from .talker import _fresh_seed
        base_seed = int(kwargs["seed"]) if kwargs.get("seed") is not None else _fresh_seed()
        for b0 in range(0, embeds.shape[0], mb):     # larger request lists run as waves of max_batch rows
            sl = slice(b0, b0 + mb)
            e, m = embeds[sl], mask[sl]
            drop = int((1 - m).sum(-1).min())        # a wave may be over-padded relative to its own longest row
            out = self.talker.generate(e[:, drop:], m[:, drop:], trailing[sl], pad, max_new_tokens=max_new_tokens,
                                       min_new_tokens=2, do_sample=do_sample, top_k=top_k, top_p=top_p,
                                       temperature=temperature, subtalker_dosample=subtalker_dosample,
                                       subtalker_top_k=subtalker_top_k, subtalker_top_p=subtalker_top_p,
                                       subtalker_temperature=subtalker_temperature,
                                       eos_token_id=eos_token_id if eos_token_id is not None else c.codec_eos_token_id,
                                       repetition_penalty=repetition_penalty, suppress_tokens=suppress,
                                       seed=base_seed + b0 // mb)
            first = out.codes[:, :, 0]
            stop = first == c.codec_eos_token_id                                                           # M:2283-2289
            for i in range(first.shape[0]):
                n_eff = int(torch.argmax(stop[i].int())) if bool(stop[i].any()) else first.shape[1]
                codes_all.append(out.codes[i, :n_eff])
                hidden_all.append(out.hidden[i, :n_eff] if out.hidden is not None else None)
        return codes_all, hidden_all


    @torch.no_grad()
    def generate_stream(self, input_ids: Optional[List[torch.Tensor]] = None, instruct_ids: Optional[List[torch.Tensor]] = None,
                        ref_ids: Optional[List[torch.Tensor]] = None, voice_clone_prompt: Optional[dict] = None,
                        languages: List[str] = None, speakers: List[str] = None, non_streaming_mode: bool = False,
                        packet_frames: int = 4, max_new_tokens: int = 4096, do_sample: bool = True, top_k: int = 50,
                        top_p: float = 1.0, temperature: float = 0.9, subtalker_dosample: bool = True, subtalker_top_k: int = 50,
                        subtalker_top_p: float = 1.0, subtalker_temperature: float = 0.9, eos_token_id: Optional[int] = None,
                        repetition_penalty: float = 1.05, **kwargs):
        """Streaming OUTPUT variant of `generate` (the reference returns whole utterances, qwen3_tts_model.py:513-515): a
        generator of packets; each packet is a list with, per request, the (k_i, G) codes it gained (k_i = 0 once the
        request hit EOS, M:2283-2289).  One wave only: len(input_ids) <= max_batch."""
        c = self.config
        if len(input_ids) > self.talker.max_batch:
            raise ValueError(f"generate_stream: {len(input_ids)} requests exceed max_batch {self.talker.max_batch}")
        embeds, mask, trailing, pad = self.assemble_prompts(input_ids, languages, speakers, instruct_ids,
                                                            non_streaming_mode, ref_ids, voice_clone_prompt)
        suppress = [i for i in range(c.vocab_size - 1024, c.vocab_size) if i != c.codec_eos_token_id]
        eos = eos_token_id if eos_token_id is not None else c.codec_eos_token_id
        alive = [True] * embeds.shape[0]
        for packet in self.talker.generate_stream(embeds, mask, trailing, pad, packet_frames=packet_frames,
                                                  max_new_tokens=max_new_tokens, min_new_tokens=2, do_sample=do_sample, top_k=top_k,
                                                  top_p=top_p, temperature=temperature, subtalker_dosample=subtalker_dosample,
                                                  subtalker_top_k=subtalker_top_k, subtalker_top_p=subtalker_top_p,
                                                  subtalker_temperature=subtalker_temperature, eos_token_id=eos,
                                                  repetition_penalty=repetition_penalty, suppress_tokens=suppress,
                                                  seed=kwargs.get("seed")):
            yield split_packet_at_eos(packet, alive, eos)
            if not any(alive):
                return


# ====================================================================================== inference wrapper
_HUB_KWARGS = ("cache_dir", "revision", "token", "local_files_only", "force_download", "proxies")


def resolve_checkpoint_dir(name_or_path: str, **kwargs) -> str:
    """The directory `from_pretrained` reads.  The reference forwards its argument to `AutoModel.from_pretrained`
    (qwen3_tts_model.py:82-121), so the examples pass hub ids ("Qwen/Qwen3-TTS-12Hz-1.7B-CustomVoice/"): a local directory is used
    as it is; anything else is resolved with `huggingface_hub.snapshot_download` (same cache, `revision` / `cache_dir` / `token` /
    `local_files_only` forwarded) when that package is importable.  A name that is neither raises OSError, as HF does -- with the
    hub's own error attached when the download was attempted (no network on a build box: a cached snapshot still resolves)."""
    path = str(name_or_path)
    if os.path.isdir(path):
        return path
    try:
        from huggingface_hub import snapshot_download
    except ImportError as e:
        raise OSError(f"{path} is not a local directory and huggingface_hub is not importable to resolve it as a hub id") from e
    repo_id = path.strip("/")
    if repo_id.count("/") > 1 or not repo_id or repo_id.startswith("."):
        raise OSError(f"{path} is neither a local directory nor a hub id of the form 'namespace/name'")
    try:
        return snapshot_download(repo_id=repo_id, **{k: kwargs[k] for k in _HUB_KWARGS if k in kwargs})
    except Exception as e:                      # offline, unknown repo, gated, ...: surface as the OSError HF's from_pretrained raises
        raise OSError(f"{path} is not a local directory and could not be resolved as a hub id ({type(e).__name__}: {e})") from e


class _TextProcessor:
    """The reference's processor is a thin wrapper over the HF Qwen2 tokenizer
    (core/models/processing_qwen3_tts.py:27); this is the same call surface on AutoTokenizer."""

    def __init__(self, path: str):
        from transformers import AutoTokenizer
        self.tok = AutoTokenizer.from_pretrained(path)

    def __call__(self, text=None, return_tensors="pt", padding=True, **kw):
        return self.tok(text, return_tensors=return_tensors, padding=padding)


MaybeList = Union[Any, List[Any]]


class Qwen3TTSModel:
    """Mirror of qwen_tts.inference.Qwen3TTSModel (qwen3_tts_model.py:54)."""

    def __init__(self, model: Qwen3TTSForConditionalGeneration, processor, generate_defaults: Optional[Dict[str, Any]] = None):
        self.model = model
        self.processor = processor
        self.generate_defaults = generate_defaults or {}
        self.device = model.device

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, **kwargs) -> "Qwen3TTSModel":
        """Same kwargs as the reference (qwen3_tts_model.py:82-121): `device_map`, `dtype`,
        `attn_implementation` (accepted; the HIP engine has one attention path)."""
        from safetensors.torch import load_file
        from .codec import Qwen3TTSTokenizer
        path = resolve_checkpoint_dir(pretrained_model_name_or_path, **kwargs)
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        sd = {}
        for fn in sorted(os.listdir(path)):
            if fn.endswith(".safetensors"):
                sd.update(load_file(os.path.join(path, fn)))
        device = str(kwargs.get("device_map", kwargs.get("device", "cuda:0")))
        dtype = kwargs.get("dtype", kwargs.get("torch_dtype", torch.bfloat16))
        gen_cfg = None
        gc_path = os.path.join(path, "generation_config.json")                # M:1922-1936
        if os.path.exists(gc_path):
            with open(gc_path) as f:
                gen_cfg = json.load(f)
        # KV capacity: the checkpoint's own `max_new_tokens` (8192 in the released generation_config.json; README default 2048)
        # plus room for the prompt, unless the caller sizes it.  288 GB of HBM makes this cheap (1.7B, batch 8: 0.9 MB per token).
        max_seq = kwargs.get("max_seq")
        if max_seq is None:
            want = int((gen_cfg or {}).get("max_new_tokens", 2048) or 2048)
            max_seq = min(16384, ((want + 1024 + 255) // 256) * 256)
        model = Qwen3TTSForConditionalGeneration(cfg, sd, device=device, dtype=dtype,
                                                 max_batch=kwargs.get("max_batch", 8), max_seq=int(max_seq))
        st_dir = os.path.join(path, "speech_tokenizer")                       # M:1900-1920
        if os.path.isdir(st_dir):
            model.load_speech_tokenizer(Qwen3TTSTokenizer.from_pretrained(st_dir, device_map=device, dtype=dtype,
                                                                          max_batch=kwargs.get("max_batch", 8)))
        if gen_cfg is not None:
            model.load_generate_config(gen_cfg)
        return cls(model=model, processor=_TextProcessor(path), generate_defaults=model.generate_config)

    # ---- helpers (qwen3_tts_model.py:123-185, 263-352)
    def _supported_languages_set(self) -> Optional[set]:
        v = self.model.get_supported_languages()
        return None if v is None else set(str(x).lower() for x in v)

    def _supported_speakers_set(self) -> Optional[set]:
        v = self.model.get_supported_speakers()
        return None if v is None else set(str(x).lower() for x in v)

    def _validate_languages(self, languages: List[str]) -> None:
        supported = self._supported_languages_set()
        if supported is None:
            return
        bad = [l for l in languages if l is None or str(l).lower() not in supported]
        if bad:
            raise ValueError(f"Unsupported languages: {bad}. Supported: {sorted(supported)}")

    def _validate_speakers(self, speakers: List[Optional[str]]) -> None:
        supported = self._supported_speakers_set()
        if supported is None:
            return
        bad = [s for s in speakers if s is not None and s != "" and str(s).lower() not in supported]
        if bad:
            raise ValueError(f"Unsupported speakers: {bad}. Supported: {sorted(supported)}")

    def _ensure_list(self, x: MaybeList) -> List[Any]:
        return x if isinstance(x, list) else [x]

    def _build_assistant_text(self, text: str) -> str:
        return f"<|im_start|>assistant\n{text}<|im_end|>\n<|im_start|>assistant\n"

    def _build_ref_text(self, text: str) -> str:
        return f"<|im_start|>assistant\n{text}<|im_end|>\n"

    def _build_instruct_text(self, instruct: str) -> str:
        return f"<|im_start|>user\n{instruct}<|im_end|>\n"

    def _tokenize_texts(self, texts: List[str]) -> List[torch.Tensor]:
        out = []
        for text in texts:
            ids = self.processor(text=text, return_tensors="pt", padding=True)["input_ids"].to(self.device)
            out.append(ids.unsqueeze(0) if ids.dim() == 1 else ids)
        return out

    def _merge_generate_kwargs(self, do_sample=None, top_k=None, top_p=None, temperature=None, repetition_penalty=None,
                               subtalker_dosample=None, subtalker_top_k=None, subtalker_top_p=None,
                               subtalker_temperature=None, max_new_tokens=None, **kwargs) -> Dict[str, Any]:
        """user value > generate_config.json > hard default (qwen3_tts_model.py:287-352)."""
        hard = dict(do_sample=True, top_k=50, top_p=1.0, temperature=0.9, repetition_penalty=1.05,
                    subtalker_dosample=True, subtalker_top_k=50, subtalker_top_p=1.0, subtalker_temperature=0.9,
                    max_new_tokens=2048)
        user = dict(do_sample=do_sample, top_k=top_k, top_p=top_p, temperature=temperature,
                    repetition_penalty=repetition_penalty, subtalker_dosample=subtalker_dosample,
                    subtalker_top_k=subtalker_top_k, subtalker_top_p=subtalker_top_p,
                    subtalker_temperature=subtalker_temperature, max_new_tokens=max_new_tokens)
        merged = dict(kwargs)
        for k, dv in hard.items():
            merged[k] = user[k] if user[k] is not None else self.generate_defaults.get(k, dv)
        return merged

    def _unsupported(self, what: str):
        m = self.model
        return ValueError(f"model with \ntokenizer_type: {m.tokenizer_type}\ntts_model_size: {m.tts_model_size}\n"
                          f"tts_model_type: {m.tts_model_type}\ndoes not support {what}, Please check Model Card or "
                          "Readme for more details.")

    def _lang_list(self, language, n):
        if isinstance(language, list):
            langs = language
        else:
            langs = [language] * n if language is not None else ["Auto"] * n
        return langs * n if len(langs) == 1 and n > 1 else langs

    def _instruct_ids(self, instructs):
        out = []
        for ins in instructs:
            out.append(None if ins is None or ins == "" else self._tokenize_texts([self._build_instruct_text(ins)])[0])
        return out

    # ---- audio inputs (qwen3_tts_model.py:167-262): same helper names; the work is in audio_io.py
    def _is_probably_base64(self, s: str) -> bool:
        from . import audio_io
        return audio_io.is_probably_base64(s)

    def _is_url(self, s: str) -> bool:
        from . import audio_io
        return audio_io.is_url(s)

    def _decode_base64_to_wav_bytes(self, b64: str) -> bytes:
        from . import audio_io
        return audio_io.decode_base64_to_wav_bytes(b64)

    def _load_audio_to_np(self, x: str):
        """IM:196-222: wav path / URL / base64 -> (mono float32 waveform, its own sample rate); no resampling here."""
        from . import audio_io
        return audio_io.load_audio_to_np(x)

    def _normalize_audio_inputs(self, audios):
        """IM:224-262: a str, an (np.ndarray, sr) tuple, or a list of those -> list of (mono float32 waveform, sr)."""
        items = audios if isinstance(audios, list) else [audios]
        out = []
        for a in items:
            if isinstance(a, str):
                out.append(self._load_audio_to_np(a))
            elif isinstance(a, tuple) and len(a) == 2 and isinstance(a[0], np.ndarray):
                out.append((a[0].astype(np.float32), int(a[1])))
            elif isinstance(a, np.ndarray):
                raise ValueError("For numpy waveform input, pass a tuple (audio, sr).")              # IM:257
            else:
                raise TypeError(f"Unsupported audio input type: {type(a)}")                          # IM:259
        return [(w.mean(axis=-1).astype(np.float32) if w.ndim > 1 else w, sr) for w, sr in out]

    # ---- voice clone (qwen3_tts_model.py:356-636)
    def create_voice_clone_prompt(self, ref_audio, ref_text=None, x_vector_only_mode=False) -> List[VoiceClonePromptItem]:
        """qwen3_tts_model.py:356-458: reference audio -> prompt items (speech codes through the tokenizer's encoder,
        x-vector through the speaker encoder).  Audio: wav path / URL / base64 string, or (waveform np.ndarray, sr);
        WAVE decoding and resampling are restated in audio_io.py (soundfile / librosa are not in this image).
        Both encoders run on the HIP engines (validated on MI355X in round 2)."""
        if self.model.tts_model_type != "base":
            raise self._unsupported("create_voice_clone_prompt")
        audios = self._ensure_list(ref_audio)
        texts = self._ensure_list(ref_text) if isinstance(ref_text, list) else [ref_text] * len(audios)
        xvecs = self._ensure_list(x_vector_only_mode) if isinstance(x_vector_only_mode, list) else [x_vector_only_mode] * len(audios)
        if len(texts) != len(audios) or len(xvecs) != len(audios):
            raise ValueError(f"Batch size mismatch: ref_audio={len(audios)}, ref_text={len(texts)}, x_vector_only_mode={len(xvecs)}")
        from . import audio_io
        normalized = self._normalize_audio_inputs(audios)
        for i, (rtext, xv) in enumerate(zip(texts, xvecs)):
            if not xv and (rtext is None or rtext == ""):
                raise ValueError(f"ref_text is required when x_vector_only_mode=False (ICL mode). Bad index={i}")
        srs = [sr for _, sr in normalized]
        if len(set(srs)) == 1:                                                                       # IM:425-431
            ref_codes = self.model.speech_tokenizer.encode([w for w, _ in normalized], sr=srs[0]).audio_codes
        else:
            ref_codes = [self.model.speech_tokenizer.encode(w, sr=sr).audio_codes[0] for w, sr in normalized]
        items = []
        spk_sr = int(self.model.speaker_encoder_sample_rate)
        wav24 = [wav if sr == spk_sr else audio_io.resample(wav, sr, spk_sr) for wav, sr in normalized]          # IM:440-444
        # the reference embeds clip by clip (IM:446-447); here clips of equal length share a launch sequence (round 6: 74 % of this
        # call was the speaker encoder running eight times) -- a row's embedding does not depend on its neighbours
        if hasattr(self.model, "extract_speaker_embeddings"):
            spk = self.model.extract_speaker_embeddings(wav24, spk_sr)
        else:                                   # (a model object without the batched entry point: `attach()`-style stand-ins)
            spk = [self.model.extract_speaker_embedding(audio=w, sr=spk_sr) for w in wav24]
        for code, rtext, xv, emb in zip(ref_codes, texts, xvecs, spk):
            items.append(VoiceClonePromptItem(ref_code=None if xv else code, ref_spk_embedding=emb,
                                              x_vector_only_mode=bool(xv), icl_mode=bool(not xv), ref_text=rtext))
        return items

    def _prompt_items_to_voice_clone_prompt(self, items: List[VoiceClonePromptItem]) -> Dict[str, Any]:
        return dict(ref_code=[it.ref_code for it in items], ref_spk_embedding=[it.ref_spk_embedding for it in items],
                    x_vector_only_mode=[it.x_vector_only_mode for it in items], icl_mode=[it.icl_mode for it in items])

    @torch.no_grad()
    def generate_voice_clone(self, text, language=None, ref_audio=None, ref_text=None, x_vector_only_mode=False,
                             voice_clone_prompt=None, non_streaming_mode: bool = False, **kwargs
                             ) -> Tuple[List[np.ndarray], int]:
        if self.model.tts_model_type != "base":
            raise self._unsupported("generate_voice_clone")
        texts = self._ensure_list(text)
        languages = self._lang_list(language, len(texts))
        if len(texts) != len(languages):
            raise ValueError(f"Batch size mismatch: text={len(texts)}, language={len(languages)}")
        self._validate_languages(languages)
        if voice_clone_prompt is None:
            if ref_audio is None:
                raise ValueError("Either `voice_clone_prompt` or `ref_audio` must be provided.")
            voice_clone_prompt = self.create_voice_clone_prompt(ref_audio=ref_audio, ref_text=ref_text,
                                                                x_vector_only_mode=x_vector_only_mode)
        if isinstance(voice_clone_prompt, list):
            items = voice_clone_prompt
            if len(items) == 1 and len(texts) > 1:
                items = items * len(texts)
            if len(items) != len(texts):
                raise ValueError(f"Batch size mismatch: prompt={len(items)}, text={len(texts)}")
            vcp = self._prompt_items_to_voice_clone_prompt(items)
            ref_texts = [it.ref_text for it in items]
        else:
            vcp, ref_texts = voice_clone_prompt, None
        input_ids = self._tokenize_texts([self._build_assistant_text(t) for t in texts])
        ref_ids = None
        if ref_texts is not None:
            ref_ids = [None if rt is None or rt == "" else self._tokenize_texts([self._build_ref_text(rt)])[0]
                       for rt in ref_texts]
        gen_kwargs = self._merge_generate_kwargs(**kwargs)
        codes_list, _ = self.model.generate(input_ids=input_ids, ref_ids=ref_ids, voice_clone_prompt=vcp,
                                            languages=languages, non_streaming_mode=non_streaming_mode, **gen_kwargs)
        ref_codes = vcp.get("ref_code", None)
        full = []
        for i, codes in enumerate(codes_list):
            if ref_codes is not None and ref_codes[i] is not None:
                full.append(torch.cat([ref_codes[i].to(codes.device), codes], dim=0))       # IM:612-618
            else:
                full.append(codes)
        wavs_all, fs = self.model.speech_tokenizer.decode([{"audio_codes": c} for c in full])
        out = []
        for i, wav in enumerate(wavs_all):
            if ref_codes is not None and ref_codes[i] is not None:
                cut = int(int(ref_codes[i].shape[0]) / max(int(full[i].shape[0]), 1) * wav.shape[0])   # IM:622-631
                out.append(wav[cut:])
            else:
                out.append(wav)
        return out, fs

    # ---- voice design (qwen3_tts_model.py:637-730)
    @torch.no_grad()
    def generate_voice_design(self, text, instruct, language=None, non_streaming_mode: bool = True, **kwargs
                              ) -> Tuple[List[np.ndarray], int]:
        if self.model.tts_model_type != "voice_design":
            raise self._unsupported("generate_voice_design")
        texts = self._ensure_list(text)
        languages = self._lang_list(language, len(texts))
        instructs = self._ensure_list(instruct)
        if len(instructs) == 1 and len(texts) > 1:
            instructs = instructs * len(texts)
        if not (len(texts) == len(languages) == len(instructs)):
            raise ValueError(f"Batch size mismatch: text={len(texts)}, language={len(languages)}, instruct={len(instructs)}")
        self._validate_languages(languages)
        input_ids = self._tokenize_texts([self._build_assistant_text(t) for t in texts])
        gen_kwargs = self._merge_generate_kwargs(**kwargs)
        codes_list, _ = self.model.generate(input_ids=input_ids, instruct_ids=self._instruct_ids(instructs),
                                            languages=languages, non_streaming_mode=non_streaming_mode, **gen_kwargs)
        return self.model.speech_tokenizer.decode([{"audio_codes": c} for c in codes_list])

    # ---- streaming output (no reference counterpart: qwen3_tts_model.py:513-515 "only simulates streaming text input")
    @torch.no_grad()
    def stream_custom_voice(self, text, speaker, language=None, instruct=None, non_streaming_mode: bool = False,
                            packet_frames: int = 4, left_context_size: int = 25, **kwargs):
        """`generate_custom_voice` as a generator of PCM packets: yields (list of np.float32 arrays, one per request --
        empty once that request has finished --, sample_rate) every `packet_frames` frames (80 ms each).  The codes come
        from `generate_stream`; each packet is decoded with `left_context_size` frames of context, i.e. exactly the
        reference's `chunked_decode(chunk_size=packet_frames, left_context_size=...)` rule applied incrementally."""
        if self.model.tts_model_type != "custom_voice":
            raise self._unsupported("stream_custom_voice")
        texts = self._ensure_list(text)
        languages = self._lang_list(language, len(texts))
        speakers = self._ensure_list(speaker)
        if len(speakers) == 1 and len(texts) > 1:
            speakers = speakers * len(texts)
        if self.model.tts_model_size in "0b6":      # 0.6B has no instruct support (IM:799-800), as in generate_custom_voice
            instruct = None
        instructs = instruct if isinstance(instruct, list) else [instruct] * len(texts)
        if len(instructs) == 1 and len(texts) > 1:
            instructs = instructs * len(texts)
        if not (len(texts) == len(languages) == len(speakers) == len(instructs)):
            raise ValueError(f"Batch size mismatch: text={len(texts)}, language={len(languages)}, speaker={len(speakers)}, instruct={len(instructs)}")
        self._validate_languages(languages)
        self._validate_speakers(speakers)
        input_ids = self._tokenize_texts([self._build_assistant_text(t) for t in texts])
        gen_kwargs = self._merge_generate_kwargs(**kwargs)
        dec = self.model.speech_tokenizer.model.decoder
        stream = dec.stream(left_context_size)
        cb = dec.config.codebook_size
        for parts in self.model.generate_stream(input_ids=input_ids, instruct_ids=self._instruct_ids(instructs), languages=languages,
                                                speakers=speakers, non_streaming_mode=non_streaming_mode,
                                                packet_frames=packet_frames, **gen_kwargs):
            k = max(int(p.shape[0]) for p in parts)
            if k == 0:
                continue
            # the codec advances in lockstep: finished / shorter rows are padded with code 0 and their samples dropped
            batch = torch.zeros(len(parts), k, parts[0].shape[-1], dtype=torch.long, device=self.device)
            for i, p in enumerate(parts):
                batch[i, : p.shape[0]] = p.clamp(min=0, max=cb - 1)
            wav = stream.push(batch.transpose(1, 2))[:, 0]
            up = wav.shape[-1] // k
            yield [wav[i, : int(p.shape[0]) * up].cpu().numpy().astype(np.float32) for i, p in enumerate(parts)], \
                int(self.model.speech_tokenizer.model.output_sample_rate)

    # ---- custom voice (qwen3_tts_model.py:732-840)
    @torch.no_grad()
    def generate_custom_voice(self, text, speaker, language=None, instruct=None, non_streaming_mode: bool = True,
                              **kwargs) -> Tuple[List[np.ndarray], int]:
        if self.model.tts_model_type != "custom_voice":
            raise self._unsupported("generate_custom_voice")
        texts = self._ensure_list(text)
        languages = self._lang_list(language, len(texts))
        speakers = self._ensure_list(speaker)
        if self.model.tts_model_size in "0b6":      # 0.6B has no instruct support (IM:799-800)
            instruct = None
        if isinstance(instruct, list):
            instructs = instruct
        else:
            instructs = [instruct] * len(texts) if instruct is not None else [""] * len(texts)
        if len(speakers) == 1 and len(texts) > 1:
            speakers = speakers * len(texts)
        if len(instructs) == 1 and len(texts) > 1:
            instructs = instructs * len(texts)
        if not (len(texts) == len(languages) == len(speakers) == len(instructs)):
            raise ValueError(f"Batch size mismatch: text={len(texts)}, language={len(languages)}, "
                             f"speaker={len(speakers)}, instruct={len(instructs)}")
        self._validate_languages(languages)
        self._validate_speakers(speakers)
        input_ids = self._tokenize_texts([self._build_assistant_text(t) for t in texts])
        gen_kwargs = self._merge_generate_kwargs(**kwargs)
        codes_list, _ = self.model.generate(input_ids=input_ids, instruct_ids=self._instruct_ids(instructs),
                                            languages=languages, speakers=speakers,
                                            non_streaming_mode=non_streaming_mode, **gen_kwargs)
        return self.model.speech_tokenizer.decode([{"audio_codes": c} for c in codes_list])

    def get_supported_speakers(self) -> Optional[List[str]]:
        s = self._supported_speakers_set()
        return None if s is None else sorted(s)

    def get_supported_languages(self) -> Optional[List[str]]:
        s = self._supported_languages_set()
        return None if s is None else sorted(s)
