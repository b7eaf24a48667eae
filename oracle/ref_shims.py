"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Import-time compatibility shims that let the UNMODIFIED reference package
(`/root/reference/qwen_tts`, pinned to transformers==4.57.3, pyproject.toml:23)
import under the transformers 5.x that is installed in this image.  Used only by
`oracle/gen_golden.py` (golden-vector generation, runs in the build container where
/root/reference exists).  Nothing here runs on the GPU box.

What is shimmed (each item is an incompatibility of the *installed libraries*, not
of the reference's arithmetic):
  1. librosa / soundfile / sox / onnxruntime / torchaudio are absent -> stub modules
     (imported at qwen_tts/inference/qwen3_tts_model.py:23-25, core/__init__.py:16-17).
  2. `@check_model_inputs()` is called with parentheses
     (tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py:499); 5.x takes the function.
  3. `ROPE_INIT_FUNCTIONS["default"]` is gone in 5.x
     (modeling_qwen3_tts.py:538,573; tokenizer v2:260).
  4. `create_causal_mask(input_embeds=..., cache_position=...)` kwargs were renamed
     (modeling_qwen3_tts.py:1097-1106,1511-1518; tokenizer v2:537-551).
"""
import importlib.machinery
import inspect
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("QTTS_REFERENCE_ROOT", "/root/reference")


def install():
    """Install the shims and return the imported reference `qwen_tts` package."""
    if "qwen_tts" in sys.modules and getattr(sys.modules["qwen_tts"], "_qtts_shimmed", False):
        return sys.modules["qwen_tts"]
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree

    import torch
    import transformers  # noqa: F401  (must be imported BEFORE the stubs exist)

    def stub(name):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        sys.modules[name] = m
        return m

    for n in ["librosa", "librosa.filters", "soundfile", "sox", "onnxruntime", "torchaudio",
              "torchaudio.compliance", "torchaudio.compliance.kaldi"]:
        stub(n)
    sys.modules["librosa.filters"].mel = lambda **k: None
    sys.modules["librosa"].filters = sys.modules["librosa.filters"]
    sys.modules["torchaudio"].compliance = sys.modules["torchaudio.compliance"]
    sys.modules["torchaudio.compliance"].kaldi = sys.modules["torchaudio.compliance.kaldi"]

    import transformers.utils.generic as G
    if not getattr(G.check_model_inputs, "_qtts", False):
        _orig = G.check_model_inputs

        def check_model_inputs(func=None, **kw):
            if func is not None:
                return _orig(func)
            return lambda g: _orig(g)
        check_model_inputs._qtts = True
        G.check_model_inputs = check_model_inputs

    import transformers.modeling_rope_utils as R

    def _default_rope(config, device=None, **kw):
        d = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        base = config.rope_theta
        inv = 1.0 / (base ** (torch.arange(0, d, 2, dtype=torch.int64).to(device=device, dtype=torch.float) / d))
        return inv, 1.0
    R.ROPE_INIT_FUNCTIONS.setdefault("default", _default_rope)

    import transformers.masking_utils as MU
    for nm in ("create_causal_mask", "create_sliding_window_causal_mask"):
        f = getattr(MU, nm)
        if getattr(f, "_qtts", False):
            continue
        ok = set(inspect.signature(f).parameters)

        def w(*a, _f=f, _ok=ok, **k):
            if "input_embeds" in k:
                k["inputs_embeds"] = k.pop("input_embeds")
            return _f(*a, **{x: y for x, y in k.items() if x in _ok})
        w._qtts = True
        setattr(MU, nm, w)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # our own drop-in package is also called `qwen_tts`; make sure the reference wins here
    for k in [k for k in sys.modules if k == "qwen_tts" or k.startswith("qwen_tts.")]:
        del sys.modules[k]
    import qwen_tts
    assert os.path.realpath(qwen_tts.__file__).startswith(os.path.realpath(REFERENCE_ROOT)), qwen_tts.__file__
    qwen_tts._qtts_shimmed = True
    return qwen_tts
