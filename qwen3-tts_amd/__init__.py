"""MI355X-native Qwen3-TTS hot path: autoregressive speech-token decoder + 12 Hz codec decoder.

Python host code (PyTorch-ROCm for device memory / streams / torch.distributed) over the C ABI of
`libqtts.so` (include/qtts.h): hand-written HIP kernels for gfx950.  The public names mirror the
reference's `qwen_tts.inference` API.
"""
from .config import CodecDecoderConfig, TalkerConfig  # noqa: F401
from ._lib import QttsError, load_library, library_path  # noqa: F401
from .codec import CodecDecoderEngine, CodecStreamDecoder, Qwen3TTSTokenizerV2Model, Qwen3TTSTokenizer  # noqa: F401
from .talker import TalkerEngine  # noqa: F401
from .model import Qwen3TTSForConditionalGeneration, Qwen3TTSModel, VoiceClonePromptItem  # noqa: F401
from .attach import attach  # noqa: F401

__all__ = ["CodecDecoderConfig", "TalkerConfig", "QttsError", "load_library", "library_path",
           "CodecDecoderEngine", "CodecStreamDecoder", "Qwen3TTSTokenizerV2Model", "Qwen3TTSTokenizer", "TalkerEngine",
           "Qwen3TTSForConditionalGeneration", "Qwen3TTSModel", "VoiceClonePromptItem", "attach"]
