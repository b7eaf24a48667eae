"""ctypes binding of libqtts.so (include/qtts.h).  The product path has NO fallback: if the HIP
library is missing or a call fails, a QttsError is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

QTTS_F32, QTTS_BF16 = 0, 1


class QttsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libqtts error {code}: {msg}")
        self.code = code


class CodecConfigC(C.Structure):
    _fields_ = [("codebook_size", C.c_int32), ("codebook_dim", C.c_int32), ("hidden_size", C.c_int32),
                ("latent_dim", C.c_int32), ("num_attention_heads", C.c_int32), ("num_key_value_heads", C.c_int32),
                ("head_dim", C.c_int32), ("sliding_window", C.c_int32), ("intermediate_size", C.c_int32),
                ("num_hidden_layers", C.c_int32), ("num_quantizers", C.c_int32), ("n_upsample_rates", C.c_int32),
                ("upsample_rates", C.c_int32 * 8), ("n_upsampling_ratios", C.c_int32),
                ("upsampling_ratios", C.c_int32 * 8), ("decoder_dim", C.c_int32), ("rms_norm_eps", C.c_float),
                ("rope_theta", C.c_float), ("compute_dtype", C.c_int32), ("max_batch", C.c_int32),
                ("max_frames", C.c_int32)]


class TalkerConfigC(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32),
                ("num_hidden_layers", C.c_int32), ("num_attention_heads", C.c_int32),
                ("num_key_value_heads", C.c_int32), ("head_dim", C.c_int32), ("rms_norm_eps", C.c_float),
                ("rope_theta", C.c_float), ("num_code_groups", C.c_int32), ("text_hidden_size", C.c_int32),
                ("codec_eos_token_id", C.c_int32), ("cp_vocab_size", C.c_int32), ("cp_hidden_size", C.c_int32),
                ("cp_intermediate_size", C.c_int32), ("cp_num_hidden_layers", C.c_int32),
                ("cp_num_attention_heads", C.c_int32), ("cp_num_key_value_heads", C.c_int32),
                ("cp_head_dim", C.c_int32), ("cp_rms_norm_eps", C.c_float), ("cp_rope_theta", C.c_float),
                ("weight_dtype", C.c_int32), ("max_batch", C.c_int32), ("max_seq", C.c_int32),
                ("use_graph", C.c_int32)]


class SamplingC(C.Structure):
    _fields_ = [("do_sample", C.c_int32), ("top_k", C.c_int32), ("top_p", C.c_float), ("temperature", C.c_float),
                ("repetition_penalty", C.c_float), ("subtalker_dosample", C.c_int32),
                ("subtalker_top_k", C.c_int32), ("subtalker_top_p", C.c_float),
                ("subtalker_temperature", C.c_float), ("seed", C.c_uint64)]


class EncoderConfigC(C.Structure):
    _fields_ = [("hidden_size", C.c_int32), ("num_filters", C.c_int32), ("num_residual_layers", C.c_int32),
                ("n_ratios", C.c_int32), ("ratios", C.c_int32 * 8), ("kernel_size", C.c_int32), ("last_kernel_size", C.c_int32),
                ("residual_kernel_size", C.c_int32), ("dilation_growth_rate", C.c_int32), ("compress", C.c_int32),
                ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32), ("num_quantizers", C.c_int32),
                ("num_semantic_quantizers", C.c_int32), ("valid_num_quantizers", C.c_int32), ("num_hidden_layers", C.c_int32),
                ("intermediate_size", C.c_int32), ("num_attention_heads", C.c_int32), ("num_key_value_heads", C.c_int32),
                ("head_dim", C.c_int32), ("sliding_window", C.c_int32), ("rope_theta", C.c_float), ("norm_eps", C.c_float),
                ("compute_dtype", C.c_int32), ("max_batch", C.c_int32), ("max_samples", C.c_int32)]


class SpeakerConfigC(C.Structure):
    _fields_ = [("mel_dim", C.c_int32), ("enc_dim", C.c_int32), ("n_blocks", C.c_int32), ("channels", C.c_int32 * 8),
                ("kernel_sizes", C.c_int32 * 8), ("dilations", C.c_int32 * 8), ("attention_channels", C.c_int32),
                ("res2net_scale", C.c_int32), ("se_channels", C.c_int32), ("n_fft", C.c_int32), ("hop_size", C.c_int32),
                ("win_size", C.c_int32), ("num_mels", C.c_int32), ("compute_dtype", C.c_int32), ("max_batch", C.c_int32),
                ("max_samples", C.c_int32)]


class TalkerStatsC(C.Structure):
    _fields_ = [("frames_run", C.c_int32), ("graph_nodes", C.c_int32), ("weight_bytes_per_frame", C.c_double),
                ("gemm_ms_last", C.c_double), ("gemm_launches_last", C.c_int64), ("long_graphs", C.c_int32),
                ("attn_nsplit_last", C.c_int32), ("attn_span_last", C.c_int32), ("cp_fused_per_step", C.c_int32),
                ("cp_fused_launches_last", C.c_int64), ("cp_fused_giveups", C.c_int32), ("cp_fused_capacity", C.c_int32),
                ("cp_fused_active", C.c_int32), ("cp_mlp_per_step", C.c_int32), ("cp_layer_per_step", C.c_int32),
                ("ks_split_per_step", C.c_int32), ("reserved0", C.c_int32)]


class CodecStatsC(C.Structure):
    _fields_ = [("graph_captures", C.c_int32), ("graph_replays", C.c_int32), ("graphs_cached", C.c_int32), ("graph_nodes_last", C.c_int32)]


class GemmClassC(C.Structure):
    _fields_ = [("stack", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("min_us", C.c_double), ("max_us", C.c_double), ("bytes_per_launch", C.c_double)]


ABI_VERSION = 12          # include/qtts.h; bumped on any signature change

# every symbol include/qtts.h declares (checked by tests/test_host_logic.py::test_abi_exports_every_declared_symbol without a GPU)
SYMBOLS = ["qtts_last_error", "qtts_abi_version", "qtts_set_option", "qtts_get_option", "qtts_codec_create", "qtts_codec_destroy", "qtts_codec_bind",
           "qtts_codec_finalize", "qtts_codec_forward", "qtts_codec_decode", "qtts_codec_forward_stage",
           "qtts_codec_stream_begin", "qtts_codec_stream_push", "qtts_codec_get_stats",
           "qtts_encoder_create", "qtts_encoder_destroy", "qtts_encoder_bind", "qtts_encoder_finalize", "qtts_encoder_frames",
           "qtts_encoder_encode",
           "qtts_speaker_create", "qtts_speaker_destroy", "qtts_speaker_bind", "qtts_speaker_finalize", "qtts_speaker_mel_frames",
           "qtts_speaker_embed",
           "qtts_talker_create", "qtts_talker_destroy", "qtts_talker_bind", "qtts_talker_finalize",
           "qtts_talker_text_projection", "qtts_talker_text_embed", "qtts_talker_assemble_rows", "qtts_talker_prefill",
           "qtts_talker_generate", "qtts_talker_stream_begin", "qtts_talker_stream_step", "qtts_talker_stream_end",
           "qtts_talker_debug_logits", "qtts_talker_debug_cp_logits", "qtts_talker_get_stats", "qtts_talker_get_gemm_profile", "qtts_talker_set_teacher",
           "qtts_talker_set_profile"]


PRODUCT_LIBRARY = os.path.join(_HERE, "libqtts.so")


def library_override():
    """Path given by QTTS_LIBRARY when it names something other than the product library, else None.  The override exists for
    build VARIANTS of the same sources (`libqtts_<variant>.so` next to the product library: A/B and measuring builds, build.py)
    and for the host-emulation build the CPU suite installs (tests/hostemu/pyshim.py).  It is never silent: `load_library`
    announces it on stderr, refuses a path that is neither of the two kinds unless QTTS_LIBRARY_OK=1 says the caller means it,
    and `bench.py` / `__graft_entry__.smoke()` refuse to run with an override at all."""
    p = os.environ.get("QTTS_LIBRARY")
    if not p or os.path.abspath(p) == os.path.abspath(PRODUCT_LIBRARY):
        return None
    return p


def library_path() -> str:
    return library_override() or PRODUCT_LIBRARY


def load_library():
    """Load libqtts.so; raises QttsError (never falls back to anything) when it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if library_override():
        import sys
        base = os.path.basename(path)
        variant = os.path.dirname(os.path.abspath(path)) == _HERE and base.startswith("libqtts_") and base.endswith(".so")
        hostemu = base.startswith("libqtts_hostemu") and os.path.basename(os.path.dirname(os.path.abspath(path))) == "hostemu"
        if not (variant or hostemu or os.environ.get("QTTS_LIBRARY_OK") == "1"):
            raise QttsError(-103, f"QTTS_LIBRARY={path} is neither a build variant of this package (qwen3-tts_amd/libqtts_<variant>.so) "
                                  "nor the test suite's host-emulation build; set QTTS_LIBRARY_OK=1 to load it anyway")
        print(f"[qtts] QTTS_LIBRARY override in effect: loading {path} instead of the product library", file=sys.stderr, flush=True)
    if not os.path.exists(path):
        raise QttsError(-100, f"{path} not found -- run `python qwen3-tts_amd/build.py` (or __graft_entry__.build())")
    lib = C.CDLL(path)
    vp, i32, i64p, f32p = C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_void_p
    lib.qtts_last_error.restype = C.c_char_p
    lib.qtts_abi_version.restype = C.c_int
    lib.qtts_set_option.argtypes = [C.c_char_p, C.c_char_p]
    lib.qtts_get_option.argtypes = [C.c_char_p, C.c_char_p, i32]
    lib.qtts_codec_create.argtypes = [C.POINTER(CodecConfigC), C.POINTER(vp)]
    lib.qtts_codec_destroy.argtypes = [vp]
    lib.qtts_codec_destroy.restype = None
    lib.qtts_codec_bind.argtypes = [vp, C.c_char_p, vp, i32, i32, i64p]
    lib.qtts_codec_finalize.argtypes = [vp]
    lib.qtts_codec_forward.argtypes = [vp, vp, i32, i32, f32p, f32p, vp]
    lib.qtts_codec_decode.argtypes = [vp, vp, i32, i32, i32, i32, f32p, i64p, vp]
    lib.qtts_codec_forward_stage.argtypes = [vp, vp, i32, i32, C.c_char_p, f32p, C.c_int64, i64p, i64p, vp]
    lib.qtts_codec_get_stats.argtypes = [vp, C.POINTER(CodecStatsC)]
    lib.qtts_codec_stream_begin.argtypes = [vp, i32]
    lib.qtts_codec_stream_push.argtypes = [vp, vp, i32, f32p, vp]
    lib.qtts_encoder_create.argtypes = [C.POINTER(EncoderConfigC), C.POINTER(vp)]
    lib.qtts_encoder_destroy.argtypes = [vp]
    lib.qtts_encoder_destroy.restype = None
    lib.qtts_encoder_bind.argtypes = [vp, C.c_char_p, vp, i32, i32, i64p]
    lib.qtts_encoder_finalize.argtypes = [vp]
    lib.qtts_encoder_frames.argtypes = [vp, C.c_int64, i64p]
    lib.qtts_encoder_encode.argtypes = [vp, f32p, i32, i32, vp, vp]
    lib.qtts_speaker_create.argtypes = [C.POINTER(SpeakerConfigC), C.POINTER(vp)]
    lib.qtts_speaker_destroy.argtypes = [vp]
    lib.qtts_speaker_destroy.restype = None
    lib.qtts_speaker_bind.argtypes = [vp, C.c_char_p, vp, i32, i32, i64p]
    lib.qtts_speaker_finalize.argtypes = [vp]
    lib.qtts_speaker_mel_frames.argtypes = [vp, C.c_int64, i64p]
    lib.qtts_speaker_embed.argtypes = [vp, f32p, i32, i32, f32p, f32p, vp]
    lib.qtts_talker_create.argtypes = [C.POINTER(TalkerConfigC), C.POINTER(vp)]
    lib.qtts_talker_destroy.argtypes = [vp]
    lib.qtts_talker_destroy.restype = None
    lib.qtts_talker_bind.argtypes = [vp, C.c_char_p, vp, i32, i32, i64p]
    lib.qtts_talker_finalize.argtypes = [vp]
    lib.qtts_talker_text_projection.argtypes = [vp, f32p, i32, f32p, vp]
    lib.qtts_talker_text_embed.argtypes = [vp, vp, i32, f32p, vp]
    lib.qtts_talker_assemble_rows.argtypes = [vp, vp, i32, f32p, i32, f32p, i32, vp, i32, f32p, vp]
    lib.qtts_talker_prefill.argtypes = [vp, f32p, i32, i32, C.POINTER(C.c_int32), f32p, i32, f32p, vp]
    lib.qtts_talker_generate.argtypes = [vp, C.POINTER(SamplingC), i32, i32, i32, C.POINTER(C.c_int32), i32, vp, vp,
                                         vp, C.POINTER(C.c_int32), vp]
    lib.qtts_talker_stream_begin.argtypes = [vp, C.POINTER(SamplingC), i32, i32, i32, C.POINTER(C.c_int32), i32, vp, vp, vp]
    lib.qtts_talker_stream_step.argtypes = [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp]
    lib.qtts_talker_stream_end.argtypes = [vp, vp, C.POINTER(C.c_int32), vp]
    lib.qtts_talker_debug_logits.argtypes = [vp, f32p, vp]
    lib.qtts_talker_debug_cp_logits.argtypes = [vp, f32p, vp]
    lib.qtts_talker_get_stats.argtypes = [vp, C.POINTER(TalkerStatsC)]
    lib.qtts_talker_get_gemm_profile.argtypes = [vp, C.POINTER(GemmClassC), i32, C.POINTER(C.c_int32)]
    lib.qtts_talker_set_teacher.argtypes = [vp, vp, i32, vp, vp, vp]
    lib.qtts_talker_set_profile.argtypes = [vp, i32]
    for s in SYMBOLS:
        if s not in ("qtts_last_error", "qtts_codec_destroy", "qtts_talker_destroy", "qtts_encoder_destroy",
                     "qtts_speaker_destroy"):
            getattr(lib, s).restype = C.c_int
    if lib.qtts_abi_version() != ABI_VERSION:
        raise QttsError(-101, "libqtts.so ABI version mismatch; rebuild")
    _LIB = lib
    return lib


def set_option(name: str, value=None, lib=None) -> None:
    """An A/B switch of the library through its C ABI (include/qtts.h qtts_set_option): measuring tools and tests only.
    value None removes the override (the switch falls back to the environment variable of the same name, then to its default)."""
    key = (id(lib) if lib is not None else 0, name)
    lib = lib or load_library()
    lib.qtts_set_option.argtypes = [C.c_char_p, C.c_char_p]
    check(lib.qtts_set_option(name.encode(), None if value is None else str(value).encode()))
    if value is None:
        _OVERRIDES.pop(key, None)
    else:
        _OVERRIDES[key] = str(value)


def get_override(name: str, lib=None):
    """The value `set_option` last gave the switch, or None when it has no override (the environment's value is not an override)."""
    return _OVERRIDES.get((id(lib) if lib is not None else 0, name))


_OVERRIDES = {}          # (library, name) -> the override in force: the C table answers "override or environment", this says which


class options:
    """`with _lib.options(QTTS_SKINNY8="0"): ...` -- switches set for the block; on exit every switch goes back to the override it had
    BEFORE the block (or to none), so blocks nest.  The table is process-global: do not open blocks from concurrent threads.
    Engine-level switches are copied into an engine when it is CREATED: build the engine inside the block."""

    def __init__(self, lib=None, **kw):
        self.lib, self.kw, self.prev = lib, kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.prev[k] = get_override(k, self.lib)
            set_option(k, v, self.lib)
        return self

    def __exit__(self, *exc):
        for k in reversed(list(self.kw)):
            set_option(k, self.prev.get(k), self.lib)
        return False


def locked(fn):
    """One engine handle per (process, device), not re-entrant (include/qtts.h): serialise callers that share an
    engine object -- the reference's demo serves from a worker pool (cli/demo.py:629)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with self._lock:
            return fn(self, *a, **k)
    return wrapper


def hip_device(device, who: str):
    """The torch device an engine lives on.  There is no CPU path: anything but a HIP device ('cuda:N' under PyTorch-ROCm)
    is refused here, for every engine class."""
    import torch
    d = torch.device(device)
    if d.type != "cuda":
        raise QttsError(-102, f"{who} requires a HIP device (torch device 'cuda:N'); there is no CPU path")
    return d


def check(rc: int):
    if rc != 0:
        raise QttsError(rc, (load_library().qtts_last_error() or b"").decode(errors="replace"))


def bind_tensor(bind_fn, handle, name: str, t):
    """Bind one torch CPU tensor (fp32 or bf16, contiguous) under `name`."""
    import torch
    t = t.detach()
    if t.device.type != "cpu":
        t = t.cpu()
    if t.dtype == torch.bfloat16:
        dt = QTTS_BF16
    else:
        t = t.to(torch.float32)
        dt = QTTS_F32
    t = t.contiguous()
    shape = (C.c_int64 * max(1, t.dim()))(*([int(s) for s in t.shape] or [1]))
    check(bind_fn(handle, name.encode(), C.c_void_p(t.data_ptr()), dt, max(1, t.dim()), shape))
