"""Host side of the autoregressive speech-token decoder (talker + code predictor) over libqtts.

`TalkerEngine.generate` mirrors the seam S2 of the reference:
`self.talker.generate(inputs_embeds, attention_mask, trailing_text_hidden, tts_pad_embed, **talker_kwargs)`
(qwen_tts/core/models/modeling_qwen3_tts.py:2272-2278), i.e. HF `_sample` around
Qwen3TTSTalkerForConditionalGeneration.forward (M:1636-1744).  All arithmetic runs in the HIP library.
"""
import ctypes as C
import threading
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch

from . import _lib
from .config import TalkerConfig


def _default_inv_freq(theta: float, head_dim: int) -> torch.Tensor:
    """HF 'default' rope init exactly as the reference's rotary modules compute it (M:538-541, 573-576)."""
    return 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(dtype=torch.float) / head_dim))


def _fresh_seed() -> int:
    """Default Philox seed of a sampling call: drawn from torch's ADVANCING default generator, so that two calls give two
    takes (as the reference's torch.multinomial does) while `torch.manual_seed(s)` still makes a run reproducible."""
    return int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item())



@dataclass
class TalkerGenerateOutput:
    codes: torch.Tensor        # (B, n_frames, G) int64, untrimmed (M:2280)
    hidden: Optional[torch.Tensor]  # (B, n_frames, H) float32 `past_hidden` per frame (M:2281)
    tokens: torch.Tensor       # (B, n_tokens) int64 sampled codebook-0 tokens (HF `sequences`)
    n_frames: int
    own: Optional[torch.Tensor] = None           # teacher forcing only: (B, F + 1, G) int32, the engine's OWN greedy choices
    logits_trace: Optional[torch.Tensor] = None  # teacher forcing only: (n_steps, B, vocab) raw cb-0 logits of `logit_steps`


_SKIP_PREFIXES = ("speaker_encoder.",)



def _as_index(v):
    """An integer-like scalar (int, numpy / torch integer) as int, else None (bool is not an index here)."""
    import operator
    if isinstance(v, bool):
        return None
    try:
        return operator.index(v)
    except TypeError:
        return None


def _check_warpers(**kw):
    """HF's own argument checks (transformers generation/logits_process.py: TopKLogitsWarper / TopPLogitsWarper /
    TemperatureLogitsWarper constructors), so that a bad value fails like the reference instead of reaching the kernel.
    `top_p` follows HF literally: only values < 0 or > 1 raise; `top_p == 0` is legal there (the sorted cumulative mass cut removes
    everything and `min_tokens_to_keep = 1` puts the top token back) and is run here as `top_k = 1` (`_resolve_top`).  Integer-like
    scalars (numpy / torch integers) are accepted for `top_k`, which is more lenient than HF's `isinstance(top_k, int)`."""
    for name in ("top_k", "subtalker_top_k"):
        v = kw.get(name)
        if v is not None and v != 0 and (_as_index(v) is None or _as_index(v) < 0):
            raise ValueError(f"`{name}` has to be a strictly positive integer, but is {v}")
    for name in ("top_p", "subtalker_top_p"):
        v = kw.get(name)
        if v is not None and not (0.0 <= float(v) <= 1.0):
            raise ValueError(f"`{name}` has to be a float > 0 and < 1, but is {v}")
    for name in ("temperature", "subtalker_temperature"):
        v = kw.get(name)
        if v is not None and not float(v) > 0.0:
            raise ValueError(f"`{name}` (={v}) has to be a strictly positive float")


def _resolve_top(top_k, top_p):
    """(top_k, top_p) as the kernel takes them: `top_p == 0` keeps exactly the top token under HF's rule (see `_check_warpers`),
    which is `top_k = 1` with no nucleus cut."""
    k = _as_index(top_k) if top_k else 0
    p = float(top_p) if top_p is not None else 1.0
    if p == 0.0:
        return 1, 1.0
    return int(k or 0), p


class TalkerEngine:
    """Owns one `qtts_talker` handle."""

    def __init__(self, config: Any, state_dict: Dict[str, torch.Tensor], weight_dtype: torch.dtype = torch.bfloat16,
                 device: str = "cuda:0", max_batch: int = 8, max_seq: int = 4096, use_graph: bool = True, shared_device: bool = False):
        """`shared_device`: this engine will run BESIDE another talker engine on the same device (a throughput job with several engines per GPU:
        bench.py --workload clone-shard, `sharding.engine_partition`).  It then keeps the decode GEMMs where an engine that has the device to
        itself runs the code predictor's MLP as one launch at batch 9..32 (csrc/cp_mlp32.hip): that launch's workgroups wait for each other, and
        behind another engine's kernels they are placed one by one and spin meanwhile -- measured on the MI355X, two engines x waves of 32:
        86.6 k tokens/s with it against 125-127 k without (profiles/r06_cp_mlp32.md); alone on the device it is 6 % faster per frame."""
        self.config = TalkerConfig.from_any(config)
        self.device = _lib.hip_device(device, "TalkerEngine")
        self.weight_dtype = weight_dtype
        self.max_batch, self.max_seq = int(max_batch), int(max_seq)
        self._lib = _lib.load_library()
        self._lock = threading.RLock()
        c = self.config
        tc = _lib.TalkerConfigC()
        for f in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                  "num_key_value_heads", "head_dim", "num_code_groups", "text_hidden_size", "codec_eos_token_id",
                  "cp_vocab_size", "cp_hidden_size", "cp_intermediate_size", "cp_num_hidden_layers",
                  "cp_num_attention_heads", "cp_num_key_value_heads", "cp_head_dim"):
            setattr(tc, f, int(getattr(c, f)))
        tc.rms_norm_eps, tc.rope_theta = float(c.rms_norm_eps), float(c.rope_theta)
        tc.cp_rms_norm_eps, tc.cp_rope_theta = float(c.cp_rms_norm_eps), float(c.cp_rope_theta)
        tc.weight_dtype = _lib.QTTS_BF16 if weight_dtype == torch.bfloat16 else _lib.QTTS_F32
        tc.max_batch, tc.max_seq, tc.use_graph = self.max_batch, self.max_seq, 1 if use_graph else 0
        self._h = C.c_void_p()
        import contextlib
        with torch.cuda.device(self.device), (_lib.options(QTTS_CP_MLP32="0") if shared_device else contextlib.nullcontext()):
            _lib.check(self._lib.qtts_talker_create(C.byref(tc), C.byref(self._h)))
            has_prefix = any(k.startswith("talker.") for k in state_dict)
            for name, t in state_dict.items():
                if has_prefix:
                    if not name.startswith("talker."):
                        continue
                    name = name[len("talker."):]
                if name.startswith(_SKIP_PREFIXES):
                    continue
                _lib.bind_tensor(self._lib.qtts_talker_bind, self._h, name, t)
            _lib.bind_tensor(self._lib.qtts_talker_bind, self._h, "model.rotary_emb.inv_freq",
                             _default_inv_freq(c.rope_theta, c.head_dim))
            _lib.bind_tensor(self._lib.qtts_talker_bind, self._h, "code_predictor.model.rotary_emb.inv_freq",
                             _default_inv_freq(c.cp_rope_theta, c.cp_head_dim))
            _lib.check(self._lib.qtts_talker_finalize(self._h))
            # hipGraph capture needs a non-default stream; everything the engine does runs on this one
            self._stream = torch.cuda.Stream(device=self.device)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.qtts_talker_destroy(h)

    def _s(self):
        return C.c_void_p(self._stream.cuda_stream)

    @_lib.locked
    def set_profile(self, enable):
        """0 / False: off.  1 / True: time every decode-GEMM launch of the real frame step on its own (see `gemm_profile`).
        2: round 2's measurement, the GEMM launches of one frame step replayed in isolation (that call yields no codes)."""
        _lib.check(self._lib.qtts_talker_set_profile(self._h, int(enable)))

    @_lib.locked
    def gemm_profile(self) -> List[dict]:
        """Per-class result of the last generate call made under `set_profile(1)` (include/qtts.h `qtts_gemm_class`)."""
        buf = (_lib.GemmClassC * 64)()
        n = C.c_int32(0)
        _lib.check(self._lib.qtts_talker_get_gemm_profile(self._h, buf, 64, C.byref(n)))
        return [{f[0]: getattr(buf[i], f[0]) for f in buf[i]._fields_} for i in range(min(64, n.value))]

    @_lib.locked
    def stats(self) -> dict:
        st = _lib.TalkerStatsC()
        _lib.check(self._lib.qtts_talker_get_stats(self._h, C.byref(st)))
        return {f[0]: getattr(st, f[0]) for f in st._fields_}

    # ------------------------------------------------------------------ text_projection (prompt assembly)
    @_lib.locked
    def text_projection(self, x: torch.Tensor) -> torch.Tensor:
        """Qwen3TTSTalkerResizeMLP (M:808-816): (..., text_hidden) -> (..., hidden), fp32."""
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).to(self.device, torch.float32).contiguous()
        y = torch.empty(x2.shape[0], self.config.hidden_size, dtype=torch.float32, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        with torch.cuda.device(self.device), torch.cuda.stream(self._stream):
            _lib.check(self._lib.qtts_talker_text_projection(self._h, C.c_void_p(x2.data_ptr()), x2.shape[0],
                                                             C.c_void_p(y.data_ptr()), self._s()))
        cur.wait_stream(self._stream)
        x2.record_stream(self._stream)
        return y.reshape(*shp[:-1], self.config.hidden_size)

    def _clamp_new_tokens(self, T: int, max_new_tokens: int) -> int:
        """HF treats `max_new_tokens` as an upper bound (generation_config.json of the released checkpoints asks for 8192);
        the engine's KV capacity is `max_seq`, fixed at construction.  A request that asks for more than fits is cut at the
        capacity (with a warning, once) instead of being refused; a prompt that leaves no room at all is an error."""
        room = self.max_seq - T
        if room < 1:
            raise ValueError(f"prompt ({T} rows) does not fit max_seq ({self.max_seq}) given at construction")
        if max_new_tokens > room:
            if not getattr(self, "_warned_clamp", False):
                import warnings
                warnings.warn(f"max_new_tokens={max_new_tokens} exceeds the KV capacity left after the prompt "
                              f"(max_seq {self.max_seq} - prompt {T} = {room}); generation is capped at {room} tokens. "
                              f"Pass a larger max_seq at construction for longer utterances.")
                self._warned_clamp = True
            max_new_tokens = room
        return max_new_tokens

    # ------------------------------------------------------------------ generate (seam S2)
    @_lib.locked
    def text_embed(self, ids: torch.Tensor) -> torch.Tensor:
        """text_projection(text_embedding[ids]) on device (M:2076-2080): ids int64 (n,) -> (n, H) fp32."""
        ids = ids.reshape(-1).to(self.device, torch.long).contiguous()
        y = torch.empty(ids.numel(), self.config.hidden_size, dtype=torch.float32, device=self.device)
        if ids.numel() == 0:
            return y
        self._stream.wait_stream(torch.cuda.current_stream(self.device))     # `ids` was produced on the caller's stream
        with torch.cuda.device(self.device), torch.cuda.stream(self._stream):
            _lib.check(self._lib.qtts_talker_text_embed(self._h, C.c_void_p(ids.data_ptr()), ids.numel(),
                                                        C.c_void_p(y.data_ptr()), self._s()))
        torch.cuda.current_stream(self.device).wait_stream(self._stream)
        return y

    @_lib.locked
    def assemble_rows(self, desc: torch.Tensor, proj: Optional[torch.Tensor] = None, spk: Optional[torch.Tensor] = None,
                      ref_codes: Optional[torch.Tensor] = None) -> torch.Tensor:
        """out[r] = proj[text_row] + codec-side term, from int32 descriptors (rows, 4) = {text_row, codec_id, spk_row,
        ref_frame} (-1 = absent).  See include/qtts.h `qtts_talker_assemble_rows`."""
        dev = self.device
        desc = desc.reshape(-1, 4).to(dev, torch.int32).contiguous()
        out = torch.empty(desc.shape[0], self.config.hidden_size, dtype=torch.float32, device=dev)
        if desc.shape[0] == 0:
            return out
        proj = None if proj is None or proj.numel() == 0 else proj.to(dev, torch.float32).contiguous()
        spk = None if spk is None or spk.numel() == 0 else spk.to(dev, torch.float32).contiguous()
        ref = None if ref_codes is None or ref_codes.numel() == 0 else ref_codes.to(dev, torch.long).contiguous()
        if ref is not None and (ref.dim() != 2 or ref.shape[1] != self.config.num_code_groups):
            raise ValueError(f"ref_codes must be (frames, {self.config.num_code_groups})")
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        self._stream.wait_stream(torch.cuda.current_stream(dev))             # desc / proj / spk / ref come from the caller's stream
        with torch.cuda.device(dev), torch.cuda.stream(self._stream):
            _lib.check(self._lib.qtts_talker_assemble_rows(
                self._h, C.c_void_p(desc.data_ptr()), desc.shape[0], ptr(proj), 0 if proj is None else proj.shape[0],
                ptr(spk), 0 if spk is None else spk.shape[0], ptr(ref), 0 if ref is None else ref.shape[0],
                C.c_void_p(out.data_ptr()), self._s()))
        torch.cuda.current_stream(dev).wait_stream(self._stream)
        return out

    @_lib.locked
    def generate(self, *args, **kw) -> TalkerGenerateOutput:
        """Seam S2 (`talker.generate`, M:2272): see `_generate_once` for the arguments.  One thing is added around it: a generation
        that ended on the fused launches' give-up flag (a consumer workgroup lost its producers -- another PROCESS on the device took
        the compute units; csrc/talker_engine.hip: check_fused_flag) has produced nothing the caller saw, and the engine has left the
        fused launches for good, so the request is re-run ONCE here on the separate launches instead of surfacing a bare error.
        (`generate_stream` cannot do that after it has yielded packets: there the error reaches the caller, whose retry runs on the
        separate launches.)"""
        giveups = self.stats()["cp_fused_giveups"]
        try:
            return self._generate_once(*args, **kw)
        except _lib.QttsError:
            if self.stats()["cp_fused_giveups"] <= giveups:
                raise
            import warnings
            warnings.warn("a fused code-predictor launch gave up waiting for its producers (device shared with another process?); "
                          "this engine now runs the separate launches and the request is re-run once")
            return self._generate_once(*args, **kw)

    def _generate_once(self, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor, trailing_text_hidden: torch.Tensor,
                 tts_pad_embed: torch.Tensor, max_new_tokens: int = 2048, min_new_tokens: int = 2,
                 do_sample: bool = True, top_k: Optional[int] = 50, top_p: Optional[float] = 1.0,
                 temperature: Optional[float] = 0.9, subtalker_dosample: bool = True,
                 subtalker_top_k: Optional[int] = 50, subtalker_top_p: Optional[float] = 1.0,
                 subtalker_temperature: Optional[float] = 0.9, eos_token_id: Optional[int] = None,
                 repetition_penalty: float = 1.05, suppress_tokens: Optional[List[int]] = None,
                 output_hidden_states: bool = True, return_dict_in_generate: bool = True,
                 seed: Optional[int] = None, teacher_codes: Optional[torch.Tensor] = None,
                 logit_steps: Optional[List[int]] = None, **unused) -> TalkerGenerateOutput:
        """`teacher_codes` (B, F, G) switches on the diagnostic teacher-forced mode (include/qtts.h `qtts_talker_set_teacher`):
        greedy, exactly F frames; the engine's own choices come back in `.own`, the raw cb-0 logits of the token steps listed in
        `logit_steps` in `.logits_trace`."""
        c = self.config
        if inputs_embeds.dim() != 3 or inputs_embeds.shape[-1] != c.hidden_size:
            raise ValueError(f"inputs_embeds must be (B, T, {c.hidden_size})")
        B, T, H = inputs_embeds.shape
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch} given at construction")
        if teacher_codes is not None:
            if teacher_codes.dim() != 3 or teacher_codes.shape[0] != B or teacher_codes.shape[2] != c.num_code_groups:
                raise ValueError(f"teacher_codes must be (B, F, {c.num_code_groups})")
            F_t = int(teacher_codes.shape[1])
            max_new_tokens = min_new_tokens = F_t + 1
            do_sample = subtalker_dosample = False
        mask = attention_mask.to("cpu", torch.long)
        if mask.shape != (B, T):
            raise ValueError("attention_mask must be (B, T)")
        n_pad = (1 - mask).sum(-1)
        # the reference only ever builds LEFT-padded masks (M:2251-2254); anything else is not this path
        expect = (torch.arange(T)[None, :] >= n_pad[:, None]).long()
        if not torch.equal(mask, expect) or int(n_pad.max()) >= T:
            raise ValueError("attention_mask must be left-padded: [0]*n_pad + [1]*(T-n_pad) per row")
        max_new_tokens = self._clamp_new_tokens(T, int(max_new_tokens))
        eos = c.codec_eos_token_id if eos_token_id is None else int(eos_token_id)
        if suppress_tokens is None:
            suppress_tokens = []
        _check_warpers(top_k=top_k, top_p=top_p, temperature=temperature, subtalker_top_k=subtalker_top_k,
                       subtalker_top_p=subtalker_top_p, subtalker_temperature=subtalker_temperature)
        sp = _lib.SamplingC()
        sp.do_sample = 1 if do_sample else 0
        sp.top_k, sp.top_p = _resolve_top(top_k, top_p)
        sp.temperature = float(temperature) if temperature is not None else 1.0
        sp.repetition_penalty = float(repetition_penalty) if repetition_penalty is not None else 1.0
        sp.subtalker_dosample = 1 if subtalker_dosample else 0
        sp.subtalker_top_k, sp.subtalker_top_p = _resolve_top(subtalker_top_k, subtalker_top_p)
        sp.subtalker_temperature = float(subtalker_temperature) if subtalker_temperature is not None else 1.0
        sp.seed = int(seed) & 0xFFFFFFFFFFFFFFFF if seed is not None else _fresh_seed()

        dev = self.device
        emb = inputs_embeds.to(dev, torch.float32).contiguous()
        trail = trailing_text_hidden.to(dev, torch.float32).contiguous()
        if trail.dim() != 3 or trail.shape[0] != B or trail.shape[2] != H or trail.shape[1] < 1:
            raise ValueError("trailing_text_hidden must be (B, Tt >= 1, H)")
        pad = tts_pad_embed.to(dev, torch.float32).reshape(-1).contiguous()
        if pad.numel() != H:
            raise ValueError("tts_pad_embed must have H elements")
        max_frames = max(1, max_new_tokens - 1)
        codes = torch.zeros(B, max_frames, c.num_code_groups, dtype=torch.int64, device=dev)
        hidden = torch.zeros(B, max_frames, H, dtype=torch.float32, device=dev) if output_hidden_states else None
        tokens = torch.full((B, max_new_tokens), -1, dtype=torch.int64, device=dev)
        npad_c = (C.c_int32 * B)(*[int(x) for x in n_pad])
        sup_c = (C.c_int32 * max(1, len(suppress_tokens)))(*[int(x) for x in suppress_tokens])
        n_frames = C.c_int32(0)
        own = trace = None
        if teacher_codes is not None:
            if F_t != max_new_tokens - 1:
                raise ValueError("teacher_codes: the forced frames do not fit max_seq")
            tc = teacher_codes.to(dev, torch.long).contiguous()
            # forced codes index embedding tables on the device: range-check them here (cb-0 < vocab, sub-codes < cp vocab)
            if int(tc.min()) < 0 or int(tc[..., 0].max()) >= c.vocab_size or (tc.shape[2] > 1 and int(tc[..., 1:].max()) >= c.cp_vocab_size):
                raise ValueError("teacher_codes: code index out of range")
            own = torch.full((B, F_t + 1, c.num_code_groups), -1, dtype=torch.int32, device=dev)
            slots = trace = None
            if logit_steps:
                sl = [-1] * (F_t + 1)
                for k, i in enumerate(logit_steps):
                    if not 0 <= int(i) <= F_t:
                        raise ValueError("logit_steps entries must be token steps in [0, F]")
                    sl[int(i)] = k
                slots = torch.tensor(sl, dtype=torch.int32, device=dev)
                trace = torch.zeros(len(logit_steps), B, c.vocab_size, dtype=torch.float32, device=dev)
        cur = torch.cuda.current_stream(dev)
        self._stream.wait_stream(cur)
        with torch.cuda.device(dev), torch.cuda.stream(self._stream):
            _lib.check(self._lib.qtts_talker_prefill(self._h, C.c_void_p(emb.data_ptr()), B, T, npad_c,
                                                     C.c_void_p(trail.data_ptr()), trail.shape[1],
                                                     C.c_void_p(pad.data_ptr()), self._s()))
            if teacher_codes is not None:
                _lib.check(self._lib.qtts_talker_set_teacher(self._h, C.c_void_p(tc.data_ptr()), F_t, C.c_void_p(own.data_ptr()),
                                                             C.c_void_p(slots.data_ptr()) if trace is not None else None,
                                                             C.c_void_p(trace.data_ptr()) if trace is not None else None))
            try:
                _lib.check(self._lib.qtts_talker_generate(
                    self._h, C.byref(sp), int(max_new_tokens), int(min_new_tokens), eos, sup_c, len(suppress_tokens),
                    C.c_void_p(codes.data_ptr()), C.c_void_p(hidden.data_ptr()) if hidden is not None else None,
                    C.c_void_p(tokens.data_ptr()), C.byref(n_frames), self._s()))
            finally:
                if teacher_codes is not None:
                    _lib.check(self._lib.qtts_talker_set_teacher(self._h, None, 0, None, None, None))
        cur.wait_stream(self._stream)
        nf = int(n_frames.value)
        return TalkerGenerateOutput(codes=codes[:, :nf], hidden=hidden[:, :nf] if hidden is not None else None,
                                    tokens=tokens[:, : nf + 1], n_frames=nf, own=own, logits_trace=trace)

    def generate_stream(self, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor, trailing_text_hidden: torch.Tensor,
                        tts_pad_embed: torch.Tensor, packet_frames: int = 4, max_new_tokens: int = 2048, min_new_tokens: int = 2,
                        do_sample: bool = True, top_k: Optional[int] = 50, top_p: Optional[float] = 1.0,
                        temperature: Optional[float] = 0.9, subtalker_dosample: bool = True,
                        subtalker_top_k: Optional[int] = 50, subtalker_top_p: Optional[float] = 1.0,
                        subtalker_temperature: Optional[float] = 0.9, eos_token_id: Optional[int] = None,
                        repetition_penalty: float = 1.05, suppress_tokens: Optional[List[int]] = None,
                        seed: Optional[int] = None, **unused):
        """Streaming OUTPUT (include/qtts.h `qtts_talker_stream_*`): a generator that yields `codes[:, f0:f1]` (B, k, G)
        int64 device tensors, k <= packet_frames, as the frames are produced; same arguments and the same frames as
        `generate`.  Closing the generator early abandons the request.  Holds the engine lock while active."""
        c = self.config
        if inputs_embeds.dim() != 3 or inputs_embeds.shape[-1] != c.hidden_size:
            raise ValueError(f"inputs_embeds must be (B, T, {c.hidden_size})")
        B, T, H = inputs_embeds.shape
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch} given at construction")
        if packet_frames < 1:
            raise ValueError("packet_frames must be >= 1")
        mask = attention_mask.to("cpu", torch.long)
        if mask.shape != (B, T):
            raise ValueError("attention_mask must be (B, T)")
        n_pad = (1 - mask).sum(-1)
        expect = (torch.arange(T)[None, :] >= n_pad[:, None]).long()
        if not torch.equal(mask, expect) or int(n_pad.max()) >= T:
            raise ValueError("attention_mask must be left-padded: [0]*n_pad + [1]*(T-n_pad) per row")
        max_new_tokens = self._clamp_new_tokens(T, int(max_new_tokens))
        eos = c.codec_eos_token_id if eos_token_id is None else int(eos_token_id)
        suppress_tokens = list(suppress_tokens or [])
        _check_warpers(top_k=top_k, top_p=top_p, temperature=temperature, subtalker_top_k=subtalker_top_k,
                       subtalker_top_p=subtalker_top_p, subtalker_temperature=subtalker_temperature)
        sp = _lib.SamplingC()
        sp.do_sample = 1 if do_sample else 0
        sp.top_k, sp.top_p = _resolve_top(top_k, top_p)
        sp.temperature = float(temperature) if temperature is not None else 1.0
        sp.repetition_penalty = float(repetition_penalty) if repetition_penalty is not None else 1.0
        sp.subtalker_dosample = 1 if subtalker_dosample else 0
        sp.subtalker_top_k, sp.subtalker_top_p = _resolve_top(subtalker_top_k, subtalker_top_p)
        sp.subtalker_temperature = float(subtalker_temperature) if subtalker_temperature is not None else 1.0
        sp.seed = int(seed) & 0xFFFFFFFFFFFFFFFF if seed is not None else _fresh_seed()
        dev = self.device
        emb = inputs_embeds.to(dev, torch.float32).contiguous()
        trail = trailing_text_hidden.to(dev, torch.float32).contiguous()
        if trail.dim() != 3 or trail.shape[0] != B or trail.shape[2] != H or trail.shape[1] < 1:
            raise ValueError("trailing_text_hidden must be (B, Tt >= 1, H)")
        pad = tts_pad_embed.to(dev, torch.float32).reshape(-1).contiguous()
        if pad.numel() != H:
            raise ValueError("tts_pad_embed must have H elements")
        codes = torch.zeros(B, max(1, max_new_tokens - 1), c.num_code_groups, dtype=torch.int64, device=dev)
        npad_c = (C.c_int32 * B)(*[int(x) for x in n_pad])
        sup_c = (C.c_int32 * max(1, len(suppress_tokens)))(*[int(x) for x in suppress_tokens])
        total, fin, nf = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        with self._lock:
            self._stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.device(dev), torch.cuda.stream(self._stream):
                _lib.check(self._lib.qtts_talker_prefill(self._h, C.c_void_p(emb.data_ptr()), B, T, npad_c,
                                                         C.c_void_p(trail.data_ptr()), trail.shape[1],
                                                         C.c_void_p(pad.data_ptr()), self._s()))
                _lib.check(self._lib.qtts_talker_stream_begin(self._h, C.byref(sp), int(max_new_tokens), int(min_new_tokens), eos,
                                                              sup_c, len(suppress_tokens), C.c_void_p(codes.data_ptr()), None,
                                                              self._s()))
            seen = 0
            try:
                while not fin.value:
                    with torch.cuda.device(dev), torch.cuda.stream(self._stream):
                        _lib.check(self._lib.qtts_talker_stream_step(self._h, int(packet_frames), C.byref(total), C.byref(fin),
                                                                     self._s()))
                    if total.value > seen:              # stream_step synchronises: these frames are final
                        yield codes[:, seen:total.value]
                        seen = total.value
            finally:
                with torch.cuda.device(dev), torch.cuda.stream(self._stream):
                    _lib.check(self._lib.qtts_talker_stream_end(self._h, None, C.byref(nf), self._s()))
                torch.cuda.current_stream(dev).wait_stream(self._stream)

    @_lib.locked
    def debug_logits(self) -> torch.Tensor:
        out = torch.empty(self.max_batch, self.config.vocab_size, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.qtts_talker_debug_logits(self._h, C.c_void_p(out.data_ptr()), self._s()))
            self._stream.synchronize()
        return out

    @_lib.locked
    def debug_cp_logits(self) -> torch.Tensor:
        """(num_code_groups - 1, max_batch, cp_vocab): the code predictor's raw logits of the last frame step that ran, every pass."""
        out = torch.empty(self.config.num_code_groups - 1, self.max_batch, self.config.cp_vocab_size, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.qtts_talker_debug_cp_logits(self._h, C.c_void_p(out.data_ptr()), self._s()))
            self._stream.synchronize()
        return out
