"""CPU: the reference's OWN example scripts, executed byte for byte against this package.

BASELINE.json's north_star asks that the engine "drops into examples/test_model_12hz_*.py unchanged".  Here the two scripts the
survey names -- /root/reference/examples/test_model_12hz_custom_voice.py and test_model_12hz_base.py -- are run with `runpy` exactly as
they lie in the reference tree (nothing is copied into this repository; the test is skipped where /root/reference does not exist, i.e.
on the GPU box).  What the test supplies is only what the scripts expect from their ENVIRONMENT:
  * `from qwen_tts import Qwen3TTSModel` resolves to the alias package at the repository root (qwen_tts/ -> qwen3_tts_amd);
  * the hub ids the scripts pass to `from_pretrained` ("Qwen/Qwen3-TTS-12Hz-1.7B-CustomVoice/", "...-Base/") are redirected -- through
    the one call the loader makes for a hub id, `huggingface_hub.snapshot_download` -- to tiny synthetic checkpoint DIRECTORIES in the
    released layout (config.json, generation_config.json, model.safetensors, tokenizer files, speech_tokenizer/);
  * `soundfile` (absent from this image) is a recording stand-in that writes RIFF/WAVE with the standard library;
  * the two reference-audio URLs of the Base script are served from memory (`urllib.request.urlopen`): there is no network;
  * "cuda:0" is the host emulation build of libqtts (tests/hostemu: the product's real C++ / HIP sources on a CPU SIMT emulator).
The scripts ask for up to 2048 new tokens; the emulator runs about one frame per second, so the synthetic checkpoints are built to
END an utterance by themselves: every projected text row carries a large constant component (text_projection's output bias) and the
EOS row of codec_head points along it, so EOS wins as soon as `min_new_tokens` allows it -- the stopping rule under test is HF's own.
"""
import dataclasses
import io
import json
import os
import runpy
import sys
import types
import wave

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXAMPLES = "/root/reference/examples"
pytestmark = pytest.mark.skipif(not os.path.isdir(EXAMPLES), reason="the reference tree is not present (GPU box): its example scripts cannot be executed")


def _forced_eos(t, w):
    """See the module docstring: + 40 / sqrt(H) on every channel of every projected text row, EOS row of codec_head = 6 / sqrt(H) x ones."""
    H = t.hidden_size
    d = np.ones(H, np.float32) / np.float32(np.sqrt(H))
    w = dict(w)
    w["text_projection.linear_fc2.bias"] = (w["text_projection.linear_fc2.bias"] + 40.0 * d).astype(np.float32)
    head = w["codec_head.weight"].copy()
    head[t.codec_eos_token_id] = 6.0 * d
    w["codec_head.weight"] = head
    return w


def _wav_bytes(x, sr=24000):
    b = io.BytesIO()
    with wave.open(b, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr)
        f.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
    return b.getvalue()


@pytest.fixture(scope="module")
def env(tmp_path_factory):
    sys.path.insert(0, os.path.join(HERE, "hostemu"))
    import pyshim
    import synth
    from ckpt_util import make_tiny_checkpoint, tiny_text_tokenizer
    from safetensors.torch import save_file
    pyshim.install()
    base = tmp_path_factory.mktemp("examples")
    t = synth.talker_tiny()
    G = t.num_code_groups
    c = synth.codec_tiny()
    c.codebook_size = t.cp_vocab_size
    cw = synth.codec_weights(c)
    tw = _forced_eos(t, synth.talker_weights(t))
    # ---- CustomVoice checkpoint: the JSON files the reference's own config classes wrote (tests/golden/ckpt_tiny), a short max_new_tokens
    cv = make_tiny_checkpoint(str(base / "cv"), os.path.join(HERE, "golden"), t, tw, c, cw)
    # ---- Base checkpoint: + speaker encoder, + the speech tokenizer's encoder
    # (128 samples per code frame instead of mimi_enc_small's 16: the speaker encoder needs > 1280 samples of reference audio, and every
    # reference frame is decoded again by the ICL path -- on the emulator a frame costs a second)
    enc = dataclasses.replace(synth.mimi_enc_small(), num_quantizers=G, encoder_valid_num_quantizers=G, upsampling_ratios=(8, 8),
                              encode_downsample_rate=128)
    spk = dataclasses.replace(synth.speaker_small(), enc_dim=t.hidden_size)
    bd = base / "base"
    os.makedirs(bd / "speech_tokenizer")
    as_t = lambda v: torch.from_numpy(np.ascontiguousarray(v))
    sd = {"talker." + k: as_t(v) for k, v in tw.items()}
    sd.update({"speaker_encoder." + k: as_t(v) for k, v in synth.speaker_weights(spk).items()})
    save_file(sd, str(bd / "model.safetensors"))
    tok_sd = {"decoder." + k: as_t(v) for k, v in cw.items()}
    tok_sd.update({"encoder." + k: as_t(v) for k, v in synth.mimi_enc_weights(enc).items()})
    save_file(tok_sd, str(bd / "speech_tokenizer" / "model.safetensors"))
    cfgd = dict(synth.cfg_dict(t), tts_model_type="base", tts_model_size="tiny", tokenizer_type="12hz", speaker_encoder_config=synth.cfg_dict(spk))
    tok_cfg = dict(synth.cfg_dict(c), encoder_config=synth.cfg_dict(enc), encoder_valid_num_quantizers=G,
                   encode_downsample_rate=enc.encode_downsample_rate, input_sample_rate=24000)
    json.dump(cfgd, open(bd / "config.json", "w"))
    json.dump(tok_cfg, open(bd / "speech_tokenizer" / "config.json", "w"))
    json.dump(json.load(open(os.path.join(cv, "generation_config.json"))), open(bd / "generation_config.json", "w"))
    tiny_text_tokenizer(t).save_pretrained(str(bd))
    yield {"cv": cv, "base": str(bd), "t": t, "c": c, "enc": enc}
    pyshim.uninstall()


def _run_example(monkeypatch, tmp_path, env, script):
    import huggingface_hub
    import urllib.request
    asked, written, fetched = [], [], []

    def fake_snapshot(repo_id, **kw):
        asked.append(repo_id)
        return env["base"] if repo_id.rstrip("/").endswith("Base") else env["cv"]
    monkeypatch.setattr(huggingface_hub, "snapshot_download", fake_snapshot)
    sf = types.ModuleType("soundfile")

    def sf_write(file, data, samplerate, *a, **k):
        data = np.asarray(data)
        written.append((os.path.basename(str(file)), data.shape, data.dtype, int(samplerate), bool(np.isfinite(data).all())))
        with open(file, "wb") as f:
            f.write(_wav_bytes(data.astype(np.float32), int(samplerate)))
    sf.write = sf_write
    monkeypatch.setitem(sys.modules, "soundfile", sf)
    g = np.random.default_rng(11)
    clips = {"clone_2.wav": _wav_bytes(g.standard_normal(1536) * 0.2), "clone_1.wav": _wav_bytes(g.standard_normal(1280) * 0.2)}      # 12 and 10 code frames (the speaker encoder needs > 4 mel frames of 256 samples) at this encoder's 16 samples per frame

    class _Resp(io.BytesIO):
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def fake_urlopen(url, *a, **k):
        fetched.append(str(url))
        return _Resp(clips[str(url).rsplit("/", 1)[-1]])
    monkeypatch.setattr(urllib.request, "urlopen", fake_urlopen)
    monkeypatch.chdir(tmp_path)
    # `qwen_tts` must resolve to THIS repository's alias package, whatever ran before in this process: tests/test_attach_reference.py imports the
    # REFERENCE's `qwen_tts` (from /root/reference, which it puts on sys.path) -- in file order it runs right before this module
    for name in [n for n in sys.modules if n == "qwen_tts" or n.startswith("qwen_tts.")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.setattr(sys, "path", [ROOT] + [q for q in sys.path if os.path.abspath(q or ".") not in (ROOT, os.path.abspath(os.path.dirname(EXAMPLES)))])
    import qwen3_tts_amd
    import qwen_tts
    assert qwen_tts.Qwen3TTSModel is qwen3_tts_amd.Qwen3TTSModel, "`from qwen_tts import Qwen3TTSModel` must resolve to this package"
    path = os.path.join(EXAMPLES, script)
    before = open(path, "rb").read()
    runpy.run_path(path, run_name="__main__")               # the file as it lies in the reference tree
    assert open(path, "rb").read() == before
    return asked, written, fetched


def test_example_custom_voice_script_runs_unchanged(monkeypatch, tmp_path, env, capsys):
    """examples/test_model_12hz_custom_voice.py:35-47 (single, with instruct) and :61-67 (batch of two, one empty instruct)."""
    asked, written, _ = _run_example(monkeypatch, tmp_path, env, "test_model_12hz_custom_voice.py")
    out = capsys.readouterr().out
    assert asked == ["Qwen/Qwen3-TTS-12Hz-1.7B-CustomVoice"]
    assert "[CustomVoice Single] time:" in out and "[CustomVoice Batch] time:" in out
    names = [w[0] for w in written]
    assert names == ["qwen3_tts_test_custom_single.wav", "qwen3_tts_test_custom_batch_0.wav", "qwen3_tts_test_custom_batch_1.wav"]
    up = env["c"].total_upsample
    for name, shape, dtype, sr, finite in written:
        assert sr == 24000 and finite and len(shape) == 1 and dtype == np.float32
        assert shape[0] % up == 0 and 1 <= shape[0] // up <= 4, (name, shape)      # EOS as soon as min_new_tokens allows: a handful of frames
        assert os.path.getsize(tmp_path / name) > 44


def test_example_base_voice_clone_script_runs_unchanged(monkeypatch, tmp_path, env, capsys):
    """examples/test_model_12hz_base.py: six call shapes x {ICL, x-vector only} -- direct `generate_voice_clone(ref_audio=URL, ref_text=...)`
    and via `create_voice_clone_prompt`, single and batched prompts (the two reference clips have different lengths: the batched encoder
    and the bucketed speaker embedding both see ragged input)."""
    asked, written, fetched = _run_example(monkeypatch, tmp_path, env, "test_model_12hz_base.py")
    out = capsys.readouterr().out
    assert asked == ["Qwen/Qwen3-TTS-12Hz-1.7B-Base"]
    assert set(u.rsplit("/", 1)[-1] for u in fetched) == {"clone_1.wav", "clone_2.wav"}
    cases = [l for l in out.splitlines() if l.startswith("[case")]
    assert len(cases) == 12 and all("sr=24000" in l for l in cases), cases
    want = []
    for tag in ("icl", "xvec_only"):
        for case, n in (("case1_promptSingle_synSingle_direct", 1), ("case1_promptSingle_synSingle_promptThenGen", 1),
                        ("case2_promptSingle_synBatch_direct", 2), ("case2_promptSingle_synBatch_promptThenGen", 2),
                        ("case3_promptBatch_synBatch_direct", 2), ("case3_promptBatch_synBatch_promptThenGen", 2)):
            want += [f"{case}_{tag}_{i}.wav" for i in range(n)]
    assert [w[0] for w in written] == want
    up = env["c"].total_upsample
    for name, shape, dtype, sr, finite in written:
        assert sr == 24000 and finite and len(shape) == 1 and dtype == np.float32
        assert shape[0] % up == 0 and 1 <= shape[0] // up <= 4, (name, shape)      # the ICL cut (IM:622-631) leaves the generated frames only
    assert os.path.isdir(tmp_path / "qwen3_tts_test_voice_clone_output_wav")
