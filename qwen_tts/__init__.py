"""`qwen_tts` import alias, so that scripts written against the reference package -- `from qwen_tts import
Qwen3TTSModel, Qwen3TTSTokenizer` (reference qwen_tts/__init__.py:21-22; examples/test_model_12hz_*.py,
examples/test_tokenizer_12hz.py) -- run on the MI355X engine unchanged when this repository is on PYTHONPATH
ahead of (or instead of) the reference.  Nothing lives here: every name is the one in `qwen3_tts_amd`."""
from qwen3_tts_amd import Qwen3TTSModel, Qwen3TTSTokenizer, VoiceClonePromptItem  # noqa: F401

__all__ = ["Qwen3TTSModel", "Qwen3TTSTokenizer", "VoiceClonePromptItem"]
