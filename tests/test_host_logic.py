"""CPU: host-side logic that needs no GPU -- C-ABI surface, configuration views, request sharding
(world_size-2 gloo), input normalisation errors of the mirrored API."""
import ctypes
import json
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(libqtts):
    hdr = open(os.path.join(ROOT, "include", "qtts.h")).read()
    declared = set(re.findall(r"\b(qtts_[a-z_]+)\s*\(", hdr))
    assert len(declared) >= 19
    lib = ctypes.CDLL(libqtts)
    for s in declared:
        assert hasattr(lib, s), f"libqtts.so does not export {s}"
    from qwen3_tts_amd import _lib
    assert declared == set(_lib.SYMBOLS), "python binding and header disagree"
    lib.qtts_abi_version.restype = ctypes.c_int
    from qwen3_tts_amd import _lib as _l
    assert lib.qtts_abi_version() == _l.ABI_VERSION


def test_product_path_fails_loudly_without_gpu_or_library(monkeypatch, tmp_path):
    from qwen3_tts_amd import _lib
    from qwen3_tts_amd.codec import CodecDecoderEngine
    from qwen3_tts_amd.talker import TalkerEngine
    with pytest.raises(_lib.QttsError):
        CodecDecoderEngine({}, {}, device="cpu")
    with pytest.raises(_lib.QttsError):
        TalkerEngine({}, {}, device="cpu")
    monkeypatch.setenv("QTTS_LIBRARY", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(_lib.QttsError):
        _lib.load_library()


def test_config_views():
    from qwen3_tts_amd.config import CodecDecoderConfig, TalkerConfig
    import synth
    c = CodecDecoderConfig.from_any({"decoder_config": {"codebook_dim": 512, "upsample_rates": [8, 5, 4, 3]},
                                     "decode_upsample_rate": 1920})
    assert c.head_dim == 64 and c.total_upsample == 1920 and c.upsample_rates == (8, 5, 4, 3)
    t = TalkerConfig.from_any({"talker_config": {"hidden_size": 2048, "head_dim": 128, "num_code_groups": 16,
                                                 "code_predictor_config": {"hidden_size": 1024, "num_hidden_layers": 5},
                                                 "spk_id": {"vivian": 3000}}, "tts_model_type": "custom_voice"})
    assert t.hidden_size == 2048 and t.cp_hidden_size == 1024 and t.head_dim == 128 and t.tts_model_type == "custom_voice"
    s = synth.talker_tiny()
    t2 = TalkerConfig.from_any(s)
    assert t2.cp_hidden_size == s.cp_hidden_size and t2.codec_eos_token_id == s.codec_eos_token_id


def test_lpt_partition_and_waves():
    from qwen3_tts_amd.sharding import lpt_partition, waves
    costs = [5, 9, 1, 7, 3, 8, 2, 6]
    parts = lpt_partition(costs, 3)
    assert sorted(sum(parts, [])) == list(range(8))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(costs)
    assert lpt_partition(costs, 3) == parts                      # deterministic
    assert lpt_partition([], 2) == [[], []]
    assert waves(list(range(10)), 4) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]


def test_gather_waveforms_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import numpy as np, torch, torch.distributed as dist
        from qwen3_tts_amd.sharding import lpt_partition, gather_waveforms
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        costs = [30, 10, 20, 40, 25]
        parts = lpt_partition(costs, world)
        mine = parts[rank]
        wavs = [np.full((100 * costs[i] + i,), float(i), np.float32) for i in mine]
        out = gather_waveforms(wavs, mine, len(costs))
        if rank == 0:
            assert [o.shape[0] for o in out] == [100 * c + i for i, c in enumerate(costs)]
            assert all((o == i).all() for i, o in enumerate(out))
            print("GATHER_OK")
        else:
            assert out is None
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29653", str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GATHER_OK" in r.stdout


def test_bench_gpus_n_refuses_to_run_fewer_ranks():
    """`python bench.py --gpus 2` on a box with < 2 GPUs must fail loudly (exit code 2), never print an n_gpus: 1 line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                       text=True, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert r.returncode == 2, r.stdout + r.stderr
    assert "FATAL" in r.stderr and "refusing" in r.stderr
    assert "{" not in r.stdout


def _bench_on_emulator(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostemu", "bench_emu.py")] + list(argv),
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_gloo_on_the_emulator():
    """`python bench.py --gpus 2` starts its own 2-rank torch.distributed job; every step ends with the request shard's
    gather on rank 0.  Here: gloo backend, tiny dims, the host-emulation build of the library -- installed by the wrapper
    tests/hostemu/bench_emu.py, which calls bench.main(device="cpu") (bench.py itself cannot load anything but the HIP
    library; the line is marked INVALID as a measurement); on an N-GPU box the same path runs nccl = RCCL."""
    j = _bench_on_emulator("--gpus", "2", "--backend", "gloo", "--model", "tiny", "--batch", "2", "--frames", "3", "--steps", "1",
                           "--warmup", "0", "--no-roofline", "--no-cpu-baseline", "--no-configs", "--talker-dtype", "f32", "--codec-dtype", "f32")
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 4 and j["backend"] == "gloo"
    assert j["gather_ms_per_step"] > 0 and "INVALID" in j and j["scaling"] == "weak"
    # what the ranks report through the process group
    assert j["ranks_seen"] == [0, 1] and len(j["ms_per_step_by_rank"]) == 2 and len(j["gather_ms_per_step_by_rank"]) == 2
    assert max(j["ms_per_step_by_rank"]) <= j["ms_per_step"] * 1.001 + 1e-6


def test_bench_api_e2e_leg_on_the_emulator():
    """bench.py's `api_e2e` leg (round 4): `Qwen3TTSModel.generate_custom_voice` from Python strings to host numpy waveforms, timed
    end to end next to the S2 step, with the parts the S2 step does not contain itemised.  Here: tiny dims on the emulator."""
    j = _bench_on_emulator("--gpus", "1", "--model", "tiny", "--batch", "2", "--frames", "3", "--steps", "1", "--warmup", "0",
                           "--no-roofline", "--no-cpu-baseline", "--no-parity-mode", "--no-configs", "--talker-dtype", "f32", "--codec-dtype", "f32")
    a = j["api_e2e"]
    assert "INVALID" in j and a["calls"] >= 1 and a["ms_per_call"] > 0 and a["frames_generated_per_row"] == 3
    assert 0 < a["frames_returned_mean"] <= 3 and set(a["not_in_s2_ms"]) and a["s2_ms_per_step"] == j["ms_per_step"]


def test_bench_configs_and_voice_clone_prompt_legs_on_the_emulator():
    """bench.py's `configs` leg (round 5: BASELINE configs 2-5 on the driver's line -- codec-only, the small model at batch 8, first packet
    at batch 32, the clone-shard job at N = 1) and its `voice_clone_prompt` leg (create_voice_clone_prompt through the real wrapper, the two
    encoders behind it), at test dims on the emulator: every sub-object is there, carries its numbers and its roofline denominator."""
    j = _bench_on_emulator("--gpus", "1", "--model", "tiny", "--batch", "2", "--frames", "3", "--steps", "1", "--warmup", "0",
                           "--no-roofline", "--no-cpu-baseline", "--no-parity-mode", "--no-api-e2e", "--talker-dtype", "f32", "--codec-dtype", "f32")
    c = j["configs"]
    assert "error" not in c, c
    assert set(c) >= {"config2_codec_only", "config3_06b_b8", "config4_first_packet_b32", "config5_clone_shard_n1", "leg_seconds"}
    assert [r["dtype"] for r in c["config2_codec_only"]["runs"]] == ["f32"] and all(r["ms_p50"] > 0 and r["roofline"]["bound"] == "mfma" for r in c["config2_codec_only"]["runs"])
    assert c["config3_06b_b8"]["value"] > 0 and c["config3_06b_b8"]["roofline"]["bound"] == "hbm" and c["config3_06b_b8"]["roofline"]["achieved"] >= 0
    assert c["config4_first_packet_b32"]["p50_ms"] > 0 and c["config4_first_packet_b32"]["p99_ms"] >= c["config4_first_packet_b32"]["p50_ms"]
    assert c["config5_clone_shard_n1"]["value"] > 0 and c["config5_clone_shard_n1"]["engines_per_gpu"] == 2 and c["config5_clone_shard_n1"]["roofline"]["achieved"] >= 0
    v = j["voice_clone_prompt"]
    assert "error" not in v, v
    assert v["runs"] and all(r["ms_per_call"] > 0 and r["encoder_batched_ms"] > 0 and r["speaker_clip_by_clip_ms"] > 0 for r in v["runs"])


def test_bench_clone_shard_strong_scaling_two_ranks_gloo_on_the_emulator():
    """BASELINE config 5 under the bench launcher (`--workload clone-shard`): a FIXED job of 6 voice-clone-shaped requests dealt
    to 2 ranks by `lpt_partition`, waves of 2, every rank's variable-length waveforms gathered on rank 0 -- strong scaling."""
    j = _bench_on_emulator("--gpus", "2", "--backend", "gloo", "--model", "tiny", "--workload", "clone-shard", "--requests", "6",
                           "--batch", "2", "--frames", "3", "--steps", "1", "--warmup", "1", "--talker-dtype", "f32",
                           "--codec-dtype", "f32")
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["requests"] == 6 and "INVALID" in j
    assert j["ranks_seen"] == [0, 1] and sum(j["requests_by_rank"]) == 6 and min(j["requests_by_rank"]) >= 2
    assert j["value"] > 0 and j["gather_ms_per_step"] > 0
    assert j["wave_batch"] == 2 and j["engines_per_gpu"] == 2          # round 4: two engines per GPU by default, LPT over (rank, engine)


def test_bench_clone_shard_defaults_are_the_fast_configuration():
    """`--workload clone-shard` with no --batch / --engines runs waves of 32 on two engines per GPU, and the
    (rank, engine) partition covers every request exactly once."""
    import importlib
    from qwen3_tts_amd import sharding
    parts = sharding.engine_partition(list(range(16, 72)) * 4, 4, 2)
    flat = sorted(i for pr in parts for pe in pr for i in pe)
    assert flat == list(range(56 * 4)) and len(parts) == 4 and all(len(pr) == 2 for pr in parts)
    loads = [sum((list(range(16, 72)) * 4)[i] for i in pe) for pr in parts for pe in pr]
    assert max(loads) / (sum(loads) / len(loads)) < 1.02
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'args.batch = 32 if args.workload == "clone-shard" else 8' in src
    assert 'args.engines = 2 if args.workload == "clone-shard" else 1' in src


def test_gather_padded_gloo_world2(tmp_path):
    script = tmp_path / "g.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from qwen3_tts_amd.sharding import gather_padded
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        wav = torch.full((3, 50), float(rank + 1))
        out = gather_padded(wav, torch.tensor([50, 40 + rank, 7]))
        if rank == 0:
            w, l = out
            assert tuple(w.shape) == (world, 3, 50) and all(bool((w[r] == r + 1).all()) for r in range(world))
            assert l.tolist() == [[50, 40 + r, 7] for r in range(world)]
            print("GATHER_PADDED_OK")
        else:
            assert out is None
        dist.destroy_process_group()
    """))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29655", str(script)],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GATHER_PADDED_OK" in r.stdout


def test_tokenizer_decode_input_errors():
    from qwen3_tts_amd.codec import Qwen3TTSTokenizer
    tk = Qwen3TTSTokenizer()
    with pytest.raises(TypeError):
        tk.decode(3.14)
    class NoEncoder:                       # a tokenizer whose checkpoint had no `encoder.*` weights
        input_sample_rate = 24000
        def encode(self, x, m, return_dict=True):
            raise NotImplementedError("no encoder weights")
    tk.model = NoEncoder()
    with pytest.raises(NotImplementedError):
        tk.encode(np.zeros(10), sr=24000)
    with pytest.raises(FileNotFoundError):
        tk.encode("no_such_ref.wav")                           # a path is opened (IT:150)
    with pytest.raises(ValueError):
        tk.encode(np.zeros(10))                                # numpy input without sr (IT:190)
    with pytest.raises(TypeError, match="Mixed input types"):
        tk.encode([np.zeros(10), "x.wav"], sr=24000)           # IT:196


def _wav_bytes(x, sr, kind="int16", channels=1):
    """A RIFF/WAVE file image for tests: PCM int16 / int24 / uint8 or IEEE float32 / float64 (plain header)."""
    import struct
    x = np.asarray(x, np.float64).reshape(-1, channels)
    if kind == "int16":
        tag, bits, raw = 1, 16, np.clip(np.round(x * 32768), -32768, 32767).astype("<i2").tobytes()
    elif kind == "uint8":
        tag, bits, raw = 1, 8, np.clip(np.round(x * 128 + 128), 0, 255).astype(np.uint8).tobytes()
    elif kind == "int24":
        v = np.clip(np.round(x * 8388608), -8388608, 8388607).astype(np.int64).reshape(-1) & 0xFFFFFF
        tag, bits, raw = 1, 24, np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], 1).astype(np.uint8).tobytes()
    elif kind == "float32":
        tag, bits, raw = 3, 32, x.astype("<f4").tobytes()
    else:
        tag, bits, raw = 3, 64, x.astype("<f8").tobytes()
    fmt = struct.pack("<HHIIHH", tag, channels, sr, sr * channels * bits // 8, channels * bits // 8, bits)
    junk = b"LIST" + struct.pack("<I", 3) + b"abc" + b"\0"                  # an odd-sized chunk before fmt: word alignment
    body = b"WAVE" + junk + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(raw)) + raw
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_audio_io_wave_decoding_base64_and_resampling(tmp_path):
    """audio_io.py restates what the reference gets from soundfile / librosa (IT:101-160, IM:188-222) for WAVE input:
    exact sample scaling for every PCM / float encoding, the base64 / data-URL / path dispatch, mono down-mix, and a
    band-limited resampler with librosa's output length (soxr parity is unpinned -- checked against the analytic signal)."""
    import base64
    import wave
    from qwen3_tts_amd import audio_io
    g = np.random.default_rng(3)
    x = (g.standard_normal(1000) * 0.3).clip(-0.99, 0.99)
    for kind, q in (("int16", 32768.0), ("int24", 8388608.0), ("uint8", 128.0)):
        got, sr = audio_io.read_wav_bytes(_wav_bytes(x, 22050, kind))
        assert sr == 22050 and got.dtype == np.float32 and got.shape == (1000,)
        assert np.array_equal(got, (np.clip(np.round(x * q), -q, q - 1) / q).astype(np.float32)), kind
    for kind in ("float32", "float64"):
        got, sr = audio_io.read_wav_bytes(_wav_bytes(x, 48000, kind))
        assert np.array_equal(got, x.astype(np.float32)) and sr == 48000
    st, _ = audio_io.read_wav_bytes(_wav_bytes(np.stack([x, -x], 1), 24000, "int16", channels=2))
    assert st.shape == (1000, 2) and np.array_equal(st[:, 0], -st[:, 1])
    # the stdlib writer's file == our reader's input; path, raw base64, data URL all reach the same samples
    path = tmp_path / "ref.wav"
    with wave.open(str(path), "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(24000)
        f.writeframes(np.clip(np.round(np.stack([x, x], 1) * 32768), -32768, 32767).astype("<i2").tobytes())
    mono, sr = audio_io.load_audio_to_np(str(path))
    want = (np.clip(np.round(x * 32768), -32768, 32767) / 32768).astype(np.float32)
    assert sr == 24000 and mono.shape == (1000,) and np.allclose(mono, want, atol=1e-7)
    b64 = base64.b64encode(path.read_bytes()).decode()
    assert np.array_equal(audio_io.load_audio_to_np("data:audio/wav;base64," + b64)[0], mono)
    # the reference's heuristic (IT:101-107): a raw string counts as base64 only if it is long and has no '/' or '\\'
    assert "/" in b64 and not audio_io.is_probably_base64(b64) and not audio_io.is_probably_base64(str(path))
    raw = base64.b64encode(_wav_bytes(np.zeros(400), 24000)).decode()
    assert "/" not in raw and audio_io.is_probably_base64(raw) and audio_io.load_audio_to_np(raw)[0].shape == (400,)
    assert audio_io.is_url("https://example.com/a.wav") and not audio_io.is_url(str(path))
    for bad in (b"", b"OggS" + b"\0" * 40, b"RIFF\x04\0\0\0WAVE"):
        with pytest.raises(ValueError):
            audio_io.read_wav_bytes(bad)
    # resampling: librosa's length rule, identity at equal rates, and the analytic band-limited answer away from the edges
    for sr_in, sr_out, n in ((16000, 24000, 4001), (44100, 24000, 9000), (48000, 24000, 5000), (22050, 24000, 3000)):
        t_in, t_out = np.arange(n) / sr_in, np.arange(int(np.ceil(n * sr_out / sr_in))) / sr_out
        sig = lambda t: 0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.3 * np.sin(2 * np.pi * 3100.0 * t + 0.7)
        y = audio_io.resample(sig(t_in).astype(np.float32), sr_in, sr_out)
        assert y.dtype == np.float32 and y.shape == t_out.shape, (sr_in, y.shape, t_out.shape)
        edge = 200
        assert np.abs(y[edge:-edge] - sig(t_out)[edge:-edge]).max() <= 2e-4, (sr_in, sr_out)
    z = g.standard_normal(100).astype(np.float32)
    assert audio_io.resample(z, 24000, 24000) is z and audio_io.resample(np.zeros(0, np.float32), 16000, 24000).shape == (0,)
    assert np.allclose(audio_io.load_audio(str(path), 24000), mono) and audio_io.load_audio(str(path), 12000).shape == (500,)


def test_tokenizer_encode_accepts_paths_and_resamples(tmp_path):
    """Qwen3TTSTokenizer.encode (IT:208-257): paths / base64 / waveforms at any rate reach the model at 24 kHz, right
    zero-padded with a mask -- checked with a stand-in model that records what it is given."""
    import wave
    from qwen3_tts_amd.codec import Qwen3TTSTokenizer
    seen = {}

    class Rec:
        input_sample_rate = 24000
        def encode(self, x, m, return_dict=True):
            seen["x"], seen["m"] = x, m
            return "codes"
    tk = Qwen3TTSTokenizer()
    tk.model = Rec()
    path = tmp_path / "a.wav"
    with wave.open(str(path), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
        f.writeframes((np.sin(np.arange(1600) * 0.05) * 20000).astype("<i2").tobytes())
    assert tk.encode(str(path)) == "codes" and seen["x"].shape == (1, 2400) and int(seen["m"].sum()) == 2400
    tk.encode([np.zeros(480, np.float32), np.ones((960, 2), np.float32)], sr=48000)
    assert seen["x"].shape == (2, 480) and seen["m"].sum(1).tolist() == [240, 480]
    tk.encode(torch.zeros(100), sr=24000)
    assert seen["x"].shape == (1, 100)


def test_model_wrapper_validation():
    """Wrapper-level validation happens before any device work, so it is testable with a stand-in model."""
    from qwen3_tts_amd.model import Qwen3TTSModel

    class M:
        device = torch.device("cpu")
        tts_model_type, tts_model_size, tokenizer_type = "custom_voice", "1b7", "12hz"
        def get_supported_languages(self): return ["auto", "chinese", "english"]
        def get_supported_speakers(self): return ["vivian", "ryan"]
    w = Qwen3TTSModel(M(), processor=None, generate_defaults={"top_k": 20})
    with pytest.raises(ValueError, match="does not support generate_voice_design"):
        w.generate_voice_design("hi", "a voice")
    with pytest.raises(ValueError, match="does not support generate_voice_clone"):
        w.generate_voice_clone("hi")
    with pytest.raises(ValueError, match="Unsupported languages"):
        w.generate_custom_voice("hi", "vivian", language="klingon")
    with pytest.raises(ValueError, match="Unsupported speakers"):
        w.generate_custom_voice("hi", "nobody", language="english")
    with pytest.raises(ValueError, match="Batch size mismatch"):
        w.generate_custom_voice(["a", "b", "c"], ["vivian", "ryan"], language="english")
    kw = w._merge_generate_kwargs(temperature=0.5, foo=1)
    assert kw["top_k"] == 20 and kw["temperature"] == 0.5 and kw["max_new_tokens"] == 2048 and kw["foo"] == 1
    assert w.get_supported_speakers() == ["ryan", "vivian"]


def test_prompt_plan_reproduces_reference_prompts(golden_dir):
    """Host half of the device prompt assembly (qwen3_tts_amd.model.build_prompt_plan): execute the integer row plan
    with a CPU stand-in for the two HIP calls (oracle text_projection + table lookups, the arithmetic
    qtts_talker_text_embed / qtts_talker_assemble_rows perform) and compare with what the REFERENCE's generate() hands
    to talker.generate in all five golden cases (custom voice / voice design / voice clone ICL + x-vector, streaming
    and non-streaming)."""
    import os
    import numpy as np
    import torch
    import synth
    import talker_ref
    from prompt_cases import CASES, load_case
    from qwen3_tts_amd.config import TalkerConfig
    from qwen3_tts_amd.model import build_prompt_plan, PLAN_PAD_ROW
    t = synth.talker_tiny()
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t).items()}
    cfg = TalkerConfig.from_any(synth.cfg_dict(t))
    g = np.load(os.path.join(golden_dir, "prompt_tiny.npz"))
    emb, H, G = w["model.codec_embedding.weight"], t.hidden_size, t.num_code_groups
    for name in CASES:
        c = load_case(g, name)
        plan = build_prompt_plan(cfg, c["ids"], c["languages"], c["speakers"], c["ins"], c["non_streaming_mode"], c["ref_ids"],
                                 c["voice_clone_prompt"])
        assert plan["desc"].dtype == np.int32 and plan["text_ids"].dtype == np.int64
        with torch.no_grad():
            proj = talker_ref.text_projection(w, w["model.text_embedding.weight"][torch.from_numpy(plan["text_ids"])])
        spk = torch.stack([x.reshape(-1).float() for x in plan["spk_vectors"]]) if plan["spk_vectors"] else None
        ref = torch.cat(plan["ref_codes"], 0) if plan["ref_codes"] else None
        rows = torch.zeros(plan["desc"].shape[0], H)
        for r, (tr, cid, sr, rf) in enumerate(plan["desc"].tolist()):
            cterm = None
            if cid >= 0:
                cterm = emb[cid]
            if sr >= 0:
                assert cterm is None
                cterm = spk[sr]
            if rf >= 0:
                assert cterm is None
                cterm = emb[ref[rf, 0]]
                for k in range(1, G):
                    cterm = cterm + w[f"code_predictor.model.codec_embedding.{k - 1}.weight"][ref[rf, k]]
            if tr >= 0:
                rows[r] = proj[tr] + cterm if cterm is not None else proj[tr]
            elif cterm is not None:
                rows[r] = cterm
        n, Tm, Tt = plan["n"], plan["Tm"], plan["Tt"]
        assert np.array_equal(plan["mask"], g[f"{name}_mask"]), name
        assert np.abs(rows[: n * Tm].reshape(n, Tm, H).numpy() - g[f"{name}_embeds"]).max() <= 1e-6, name
        assert np.abs(rows[n * Tm:].reshape(n, Tt, H).numpy() - g[f"{name}_trailing"]).max() <= 1e-6, name
        assert np.abs(proj[PLAN_PAD_ROW].numpy() - g[f"{name}_tts_pad"].reshape(-1)).max() <= 1e-6, name
    import pytest
    with pytest.raises(NotImplementedError):
        build_prompt_plan(cfg, [torch.from_numpy(g["cv_ns_ids0"])], ["klingon"], ["vivian"])
    with pytest.raises(NotImplementedError):
        build_prompt_plan(cfg, [torch.from_numpy(g["cv_ns_ids0"])], ["english"], ["nobody"])


def test_checkpoint_directory_configs_and_no_cpu_path(golden_dir, tmp_path):
    """`from_pretrained` inputs without a GPU: the reference-written config.json files parse to the right dimensions,
    the HF tokenizer in the directory produces the id layout the prompt slicing assumes, and loading onto a CPU
    device fails loudly (there is no CPU product path)."""
    import json
    import os
    import pytest
    import synth
    from ckpt_util import make_tiny_checkpoint
    from qwen3_tts_amd import Qwen3TTSModel, _lib
    from qwen3_tts_amd.config import CodecDecoderConfig, TalkerConfig
    from qwen3_tts_amd.model import _TextProcessor
    t = synth.talker_tiny()
    c = synth.codec_tiny()
    c.codebook_size = t.cp_vocab_size
    path = make_tiny_checkpoint(str(tmp_path / "ckpt"), golden_dir, t, synth.talker_weights(t), c, synth.codec_weights(c))
    with open(os.path.join(path, "config.json")) as f:
        tc = TalkerConfig.from_any(json.load(f))
    want = synth.cfg_dict(t)
    for k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads",
              "head_dim", "num_code_groups", "text_hidden_size", "text_vocab_size", "codec_eos_token_id", "codec_pad_id",
              "codec_bos_id", "cp_vocab_size", "cp_hidden_size", "cp_intermediate_size", "cp_num_hidden_layers",
              "cp_num_attention_heads", "cp_num_key_value_heads", "cp_head_dim", "im_start_token_id", "tts_pad_token_id"):
        assert getattr(tc, k) == want[k], k
    assert tc.spk_id == t.spk_id and tc.codec_language_id == t.codec_language_id and tc.tts_model_type == "custom_voice"
    with open(os.path.join(path, "speech_tokenizer", "config.json")) as f:
        cc = CodecDecoderConfig.from_any(json.load(f))
    for k, v in synth.cfg_dict(c).items():
        got = getattr(cc, k)
        assert (tuple(got) == tuple(v)) if isinstance(v, (tuple, list)) else (got == v), k
    ids = _TextProcessor(path)("<|im_start|>assistant\nhi<|im_end|>\n<|im_start|>assistant\n")["input_ids"][0].tolist()
    assert ids[:3] == [t.im_start_token_id, 77, 198] and ids[-5:] == [t.im_end_token_id, 198, t.im_start_token_id, 77, 198]
    assert len(ids) == 3 + 2 + 5
    with pytest.raises(_lib.QttsError):
        Qwen3TTSModel.from_pretrained(path, device_map="cpu")


def test_engine_calls_are_serialised_per_handle():
    """`_lib.locked`: the C handles are not re-entrant, so every public engine method holds the engine's lock."""
    import threading
    import time
    from qwen3_tts_amd import _lib
    from qwen3_tts_amd.codec import CodecDecoderEngine
    from qwen3_tts_amd.talker import TalkerEngine

    class Dummy:
        def __init__(self):
            self._lock = threading.RLock()
            self.inside = 0
            self.overlap = False

        @_lib.locked
        def call(self):
            self.inside += 1
            self.overlap = self.overlap or self.inside > 1
            time.sleep(0.005)
            self.nested()                      # re-entrant from the same thread
            self.inside -= 1

        @_lib.locked
        def nested(self):
            return 1

    d = Dummy()
    th = [threading.Thread(target=lambda: [d.call() for _ in range(5)]) for _ in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not d.overlap
    for cls, names in ((TalkerEngine, ("generate", "text_embed", "assemble_rows", "text_projection")),
                       (CodecDecoderEngine, ("forward", "decode_padded", "forward_stage"))):
        for n in names:
            assert hasattr(getattr(cls, n), "__wrapped__"), f"{cls.__name__}.{n} is not serialised"


def test_stream_decoder_equals_reference_chunked_decode(golden_dir):
    """CodecStreamDecoder (host logic) driven by the ORACLE decoder: packets of k frames == the reference's
    chunked_decode(chunk_size=k, left_context_size=L) (tokenizer v2:886-896), for several (k, L) incl. L = 0 and
    T not a multiple of k; ragged packet sizes == the same rule applied packet by packet."""
    import numpy as np
    import torch
    import codec_ref
    import synth
    from qwen3_tts_amd.codec import CodecStreamDecoder
    c = synth.codec_tiny()
    w = {k: torch.from_numpy(v) for k, v in synth.codec_weights(c).items()}
    codes = torch.from_numpy(np.random.default_rng(3).integers(0, c.codebook_size, (2, c.num_quantizers, 11)))
    fwd = lambda x: codec_ref.decoder_forward(w, c, x)
    with torch.no_grad():
        for k, L in ((4, 3), (3, 0), (5, 25), (11, 2)):
            sd = CodecStreamDecoder(fwd, c.total_upsample, L)
            got = torch.cat([sd.push(codes[..., i:i + k]) for i in range(0, 11, k)], dim=-1)
            ref = codec_ref.chunked_decode(w, c, codes, chunk_size=k, left_context_size=L)
            assert got.shape == ref.shape == (2, 1, 11 * c.total_upsample)
            assert torch.equal(got, ref), (k, L)
        sd = CodecStreamDecoder(fwd, c.total_upsample, 3)
        cuts = [0, 2, 7, 8, 11]
        got = torch.cat([sd.push(codes[..., a:b]) for a, b in zip(cuts[:-1], cuts[1:])], dim=-1)
        parts = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            ctx = 3 if a - 3 > 0 else a
            parts.append(fwd(codes[..., a - ctx:b])[..., ctx * c.total_upsample:])
        assert torch.equal(got, torch.cat(parts, dim=-1))
        sd.reset()
        assert torch.equal(sd.push(codes[..., :4]), fwd(codes[..., :4]))


def test_prompt_plan_randomised_against_oracle():
    """Differential test of the host row plan over many random batches (mode x streaming x instruct x speaker x language
    x ICL / x-vector, random lengths): executing the plan with CPU stand-ins for the two HIP calls must reproduce the
    oracle's `assemble_prompts` (which is itself pinned to the reference in tests/test_oracle_golden.py)."""
    import numpy as np
    import torch
    import synth
    import talker_ref
    from qwen3_tts_amd.config import TalkerConfig
    from qwen3_tts_amd.model import build_prompt_plan, PLAN_PAD_ROW
    t = synth.talker_tiny()
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t).items()}
    cfg = TalkerConfig.from_any(synth.cfg_dict(t))
    emb, H, G = w["model.codec_embedding.weight"], t.hidden_size, t.num_code_groups
    rng = np.random.default_rng(77)

    def ids(n_body):
        body = rng.integers(0, 490, n_body).tolist()
        return torch.tensor([[t.im_start_token_id, 77, 198] + body + [t.im_end_token_id, 198, t.im_start_token_id, 77, 198]])

    for trial in range(40):
        n = int(rng.integers(1, 5))
        mode = ["custom", "design", "clone"][trial % 3]
        ns = bool(rng.integers(0, 2))
        input_ids = [ids(int(rng.integers(1, 12))) for _ in range(n)]
        langs = [["chinese", "english", "auto"][int(rng.integers(0, 3))] for _ in range(n)]
        spk = ins = ref_ids = vcp = None
        if mode == "custom":
            spk = [["vivian", "ryan"][int(rng.integers(0, 2))] for _ in range(n)]
            ins = [torch.tensor([[t.im_start_token_id, 78, 198] + rng.integers(0, 490, int(rng.integers(1, 6))).tolist() +
                                 [t.im_end_token_id, 198]]) if rng.integers(0, 2) else None for _ in range(n)]
        elif mode == "design":
            ins = [torch.tensor([[t.im_start_token_id, 78, 198] + rng.integers(0, 490, int(rng.integers(1, 6))).tolist() +
                                 [t.im_end_token_id, 198]]) for _ in range(n)]
        else:
            icl = [bool(rng.integers(0, 2)) for _ in range(n)]
            ref_ids = [torch.tensor([[t.im_start_token_id, 77, 198] + rng.integers(0, 490, int(rng.integers(1, 9))).tolist() +
                                     [t.im_end_token_id, 198]]) for _ in range(n)]
            def ref_code():
                m = int(rng.integers(1, 9))
                return torch.from_numpy(np.concatenate([rng.integers(0, t.vocab_size - 1024, (m, 1)),
                                                        rng.integers(0, t.cp_vocab_size, (m, G - 1))], 1))
            vcp = dict(ref_code=[ref_code() if icl[i] else None for i in range(n)],
                       ref_spk_embedding=[torch.from_numpy(rng.standard_normal(H).astype(np.float32)) for _ in range(n)],
                       x_vector_only_mode=[not x for x in icl], icl_mode=icl)
        with torch.no_grad():
            e0, m0, tr0, pad0 = talker_ref.assemble_prompts(w, t, input_ids, langs, spk, ins, ns, ref_ids, vcp)
            plan = build_prompt_plan(cfg, input_ids, langs, spk, ins, ns, ref_ids, vcp)
            proj = talker_ref.text_projection(w, w["model.text_embedding.weight"][torch.from_numpy(plan["text_ids"])])
        spkm = torch.stack([x.reshape(-1).float() for x in plan["spk_vectors"]]) if plan["spk_vectors"] else None
        ref = torch.cat(plan["ref_codes"], 0) if plan["ref_codes"] else None
        rows = torch.zeros(plan["desc"].shape[0], H)
        for r, (tr, cid, sr, rf) in enumerate(plan["desc"].tolist()):
            cterm = None
            if cid >= 0:
                cterm = emb[cid]
            if sr >= 0:
                cterm = spkm[sr]
            if rf >= 0:
                cterm = emb[ref[rf, 0]]
                for k in range(1, G):
                    cterm = cterm + w[f"code_predictor.model.codec_embedding.{k - 1}.weight"][ref[rf, k]]
            if tr >= 0:
                rows[r] = proj[tr] + cterm if cterm is not None else proj[tr]
            elif cterm is not None:
                rows[r] = cterm
        nn, Tm, Tt = plan["n"], plan["Tm"], plan["Tt"]
        tag = (trial, mode, ns)
        assert np.array_equal(plan["mask"], m0.numpy()), tag
        assert (rows[: nn * Tm].reshape(nn, Tm, H) - e0).abs().max() <= 1e-6, tag
        assert (rows[nn * Tm:].reshape(nn, Tt, H) - tr0).abs().max() <= 1e-6, tag
        assert (proj[PLAN_PAD_ROW] - pad0.reshape(-1)).abs().max() <= 1e-6, tag


def test_encoder_config_from_reference_tokenizer_json(golden_dir):
    """CodecEncoderConfig reads the Mimi `encoder_config` the reference's tokenizer config serialises
    (tests/golden/ckpt_tiny/speech_tokenizer/config.json, written by Qwen3TTSTokenizerV2Config)."""
    import json
    import os
    import synth
    from qwen3_tts_amd.config import CodecEncoderConfig
    with open(os.path.join(golden_dir, "ckpt_tiny", "speech_tokenizer", "config.json")) as f:
        cfg = CodecEncoderConfig.from_any(json.load(f))
    real = synth.mimi_enc_real()
    for k in ("hidden_size", "num_filters", "num_residual_layers", "kernel_size", "last_kernel_size", "residual_kernel_size",
              "dilation_growth_rate", "compress", "codebook_size", "codebook_dim", "num_quantizers", "num_semantic_quantizers",
              "num_hidden_layers", "intermediate_size", "num_attention_heads", "num_key_value_heads", "head_dim",
              "sliding_window", "encoder_valid_num_quantizers", "encode_downsample_rate"):
        assert getattr(cfg, k) == getattr(real, k), k
    assert tuple(cfg.upsampling_ratios) == tuple(real.upsampling_ratios) and abs(cfg.rope_theta - real.rope_theta) < 1e-3
    small = CodecEncoderConfig.from_any(synth.cfg_dict(synth.mimi_enc_small()))
    assert small.head_dim == 64 and small.upsampling_ratios == (4, 2) and small.encoder_valid_num_quantizers == 4


def test_create_voice_clone_prompt_wrapper_logic():
    """qwen3_tts_model.py:356-458 mirrored: batching rules, ICL needs ref_text, error types, item fields -- with stand-ins
    for the two device engines (the wrapper logic runs before / around them)."""
    import numpy as np
    import pytest
    import torch
    from qwen3_tts_amd.model import Qwen3TTSModel, VoiceClonePromptItem

    class Tok:
        def encode(self, wavs, sr=None):
            wavs = wavs if isinstance(wavs, list) else [wavs]
            self.last_sr = sr
            return type("O", (), {"audio_codes": [torch.full((max(1, len(w) * 24000 // sr // 1920), 16), i) for i, w in enumerate(wavs)]})()

    class M:
        device = torch.device("cpu")
        tts_model_type, tts_model_size, tokenizer_type = "base", "1b7", "12hz"
        speaker_encoder_sample_rate = 24000
        speech_tokenizer = Tok()
        def get_supported_languages(self): return None
        def get_supported_speakers(self): return None
        def extract_speaker_embedding(self, audio, sr):
            assert sr == 24000
            return torch.full((8,), float(len(audio)))

    w = Qwen3TTSModel(M(), processor=None, generate_defaults={})
    a, b = (np.zeros(4000, np.float32), 24000), (np.zeros(8000, np.float32), 24000)
    items = w.create_voice_clone_prompt([a, b], ref_text=["hello", "world"])
    assert all(isinstance(i, VoiceClonePromptItem) for i in items) and [i.icl_mode for i in items] == [True, True]
    assert items[1].ref_code.shape == (4, 16) and float(items[1].ref_spk_embedding[0]) == 8000.0 and items[0].ref_text == "hello"
    xv = w.create_voice_clone_prompt(a, x_vector_only_mode=True)
    assert len(xv) == 1 and xv[0].ref_code is None and xv[0].x_vector_only_mode and not xv[0].icl_mode
    with pytest.raises(ValueError, match="ref_text is required"):
        w.create_voice_clone_prompt(a)
    with pytest.raises(ValueError, match="Batch size mismatch"):
        w.create_voice_clone_prompt([a, b], ref_text=["only one"])
    with pytest.raises(FileNotFoundError):
        w.create_voice_clone_prompt("no_such_ref.wav", ref_text="x")                   # a path is opened (IM:218)
    # a 16 kHz reference: codes from the tokenizer at the ORIGINAL rate (it resamples itself, IM:425-431), the speaker
    # encoder gets a 24 kHz copy (IM:440-444); mixed rates are encoded one by one
    it16 = w.create_voice_clone_prompt((np.zeros(3200, np.float32), 16000), ref_text="x")
    assert M.speech_tokenizer.last_sr == 16000 and float(it16[0].ref_spk_embedding[0]) == 4800.0
    mixed = w.create_voice_clone_prompt([a, (np.zeros(3200, np.float32), 16000)], ref_text=["p", "q"])
    assert len(mixed) == 2 and float(mixed[1].ref_spk_embedding[0]) == 4800.0 and mixed[1].ref_code.shape == (2, 16)
    with pytest.raises(ValueError, match="pass a tuple"):
        w.create_voice_clone_prompt(np.zeros(100, np.float32), ref_text="x")            # IM:257
    with pytest.raises(TypeError):
        w.create_voice_clone_prompt(3.5, ref_text="x")                                  # IM:259
    M.tts_model_type = "custom_voice"
    with pytest.raises(ValueError, match="does not support create_voice_clone_prompt"):
        w.create_voice_clone_prompt(a, ref_text="x")
    M.tts_model_type = "base"


def test_streaming_output_host_logic():
    """Streaming output wrappers: the EOS packet split (streaming form of M:2283-2289) and `stream_custom_voice`'s packet
    assembly -- ragged rows padded for the lock-step codec, finished requests yielding empty arrays -- with stand-ins for the
    two engines."""
    import numpy as np
    import torch
    from qwen3_tts_amd.codec import CodecStreamDecoder
    from qwen3_tts_amd.model import Qwen3TTSModel, split_packet_at_eos
    alive = [True, True, False]
    pk = torch.tensor([[[5, 1], [9, 2], [7, 3]], [[1, 1], [2, 2], [3, 3]], [[9, 9], [9, 9], [9, 9]]])
    parts = split_packet_at_eos(pk, alive, 9)
    assert [p.shape[0] for p in parts] == [1, 3, 0] and alive == [False, True, False]
    assert [p.shape[0] for p in split_packet_at_eos(pk, alive, 9)] == [0, 3, 0]          # a finished row stays finished

    UP = 4

    class Dec:                                   # decoder stand-in: sample value = first code of its frame
        config = type("C", (), {"codebook_size": 50})()
        def forward(self, codes):                # (B, Q, T) -> (B, 1, T*UP)
            return codes[:, 0, :].float().repeat_interleave(UP, dim=-1).unsqueeze(1)
        def stream(self, left):
            return CodecStreamDecoder(self.forward, UP, left)

    class Tok:
        model = type("M", (), {"decoder": Dec(), "output_sample_rate": 24000})()

    class M:
        device = torch.device("cpu")
        tts_model_type, tts_model_size, tokenizer_type = "custom_voice", "1b7", "12hz"
        speech_tokenizer = Tok()
        def get_supported_languages(self): return None
        def get_supported_speakers(self): return None
        def generate_stream(self, **kw):
            assert kw["packet_frames"] == 2 and kw["speakers"] == ["a", "b"]
            yield [torch.tensor([[3, 0], [4, 0]]), torch.tensor([[7, 0], [8, 0]])]
            yield [torch.tensor([[5, 0]]), torch.tensor([[9, 0], [60, 0]])]            # row 0 ends mid-packet; 60 clamps to 49
            yield [torch.zeros(0, 2, dtype=torch.long), torch.zeros(0, 2, dtype=torch.long)]   # nothing new: no packet

    class Proc:
        def __call__(self, text=None, return_tensors="pt", padding=True):
            return {"input_ids": torch.tensor([[1, 2, 3, 4, 5, 6, 7, 8, 9, 10]])}

    w = Qwen3TTSModel(M(), Proc(), generate_defaults={})
    out = list(w.stream_custom_voice(["x", "y"], ["a", "b"], language="auto", packet_frames=2, left_context_size=1))
    assert len(out) == 2 and all(sr == 24000 for _, sr in out)
    (p0, _), (p1, _) = out
    assert [a.tolist() for a in p0] == [[3.0] * UP + [4.0] * UP, [7.0] * UP + [8.0] * UP]
    assert [a.tolist() for a in p1] == [[5.0] * UP, [9.0] * UP + [49.0] * UP] and p1[0].dtype == np.float32


def test_api_surface_matches_reference(golden_dir):
    """The drop-in boundary: every method of the reference's `Qwen3TTSModel` / `Qwen3TTSTokenizer` /
    `VoiceClonePromptItem` exists on the mirrored class with the same parameter names, order, defaults and kind.
    The expectation is tests/golden/api_surface.json, read from the reference sources by oracle/gen_api_surface.py."""
    import inspect
    import json
    import qwen3_tts_amd as pkg
    with open(os.path.join(golden_dir, "api_surface.json")) as f:
        ref = json.load(f)["classes"]
    assert set(ref) == {"Qwen3TTSModel", "Qwen3TTSTokenizer", "VoiceClonePromptItem"}
    problems = []
    for cname, body in ref.items():
        cls = getattr(pkg, cname)
        have_fields = list(getattr(cls, "__dataclass_fields__", {}))
        if have_fields[:len(body["fields"])] != body["fields"]:
            problems.append(f"{cname}: fields {have_fields} != {body['fields']}")
        for mname, m in body["methods"].items():
            raw = inspect.getattr_static(cls, mname, None)
            if raw is None:
                problems.append(f"{cname}.{mname}: missing")
                continue
            kind = "classmethod" if isinstance(raw, classmethod) else "staticmethod" if isinstance(raw, staticmethod) \
                else "property" if isinstance(raw, property) else "method"
            if kind != m["kind"]:
                problems.append(f"{cname}.{mname}: kind {kind} != {m['kind']}")
                continue
            fn = raw.__func__ if kind in ("classmethod", "staticmethod") else raw.fget if kind == "property" else raw
            fn = inspect.unwrap(fn)                   # through @torch.no_grad() etc.
            got = []
            for p in inspect.signature(fn).parameters.values():
                name = ("*" if p.kind is p.VAR_POSITIONAL else "**" if p.kind is p.VAR_KEYWORD else "") + p.name
                got.append([name, None if p.default is p.empty else repr(p.default)])
            if got != m["params"]:
                problems.append(f"{cname}.{mname}: {got} != {m['params']}")
    assert not problems, "\n".join(problems)


def test_qwen_tts_alias_exports_the_mirrored_classes():
    """`from qwen_tts import Qwen3TTSModel, Qwen3TTSTokenizer` (the reference examples' import line) resolves to the
    MI355X classes when the repository root is on the path."""
    code = ("import sys; sys.path.insert(0, %r); import qwen_tts, qwen3_tts_amd as a; "
            "assert qwen_tts.Qwen3TTSModel is a.Qwen3TTSModel and qwen_tts.Qwen3TTSTokenizer is a.Qwen3TTSTokenizer "
            "and qwen_tts.VoiceClonePromptItem is a.VoiceClonePromptItem; print('ok')" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_no_kernel_spills_or_scratch(libqtts):
    """Static check the host emulator cannot make: no gfx950 kernel of libqtts.so spills registers, uses scratch
    memory or declares more static LDS than a CU has (tools/kernel_resources.py reads the code objects' metadata)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "--so", libqtts, "--check"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert re.search(r"^(\d+) kernels; 0 with spills", out.stdout, re.M), out.stdout[-500:]
    assert int(re.search(r"^(\d+) kernels;", out.stdout, re.M).group(1)) >= 60


def test_build_variants_are_opt_in_only(libqtts):
    """The product library is the flag-free build: variants (A/B material) get their own file names, carry at least one
    -DQTTS_ flag each (`pk`: a code-generation flag), and nothing in the default flag set or in __graft_entry__.build() selects one."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("qtts_build_t", os.path.join(ROOT, "qwen3-tts_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert not any(f.startswith("-DQTTS_") for f in m.FLAGS)
    assert os.path.basename(m.OUT) == "libqtts.so"
    for name, flags in m.VARIANTS.items():
        if name == "pk":      # (round 5: the diagnosis build WITH packed fp32 math, a code-generation flag the product build turns off)
            assert flags == ["-Xclang", "-target-feature", "-Xclang", "+packed-fp32-ops"] and "-packed-fp32-ops" in m.FLAGS
            continue
        assert flags and all(f.startswith("-DQTTS_") for f in flags), name
        assert os.path.basename(m.variant_path(name)) == f"libqtts_{name}.so"
    with pytest.raises(ValueError):
        m.build(variant="no_such_variant")
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "variant" not in src
    from qwen3_tts_amd import _lib
    if "QTTS_LIBRARY" not in os.environ:
        assert os.path.basename(_lib.library_path()) == "libqtts.so"
    # measuring code stays out of the product library: the hand-off probe's entry point is exported by its own variant only
    import ctypes
    prod = ctypes.CDLL(libqtts) if isinstance(libqtts, str) else None
    assert prod is None or not hasattr(prod, "qtts_debug_persist_layer")
    assert "persist_probe.hip" not in m.SOURCES and "persist_probe.hip" in m.VARIANT_SOURCES["probe"]
    # every -D name used by a variant is defaulted to 0 in the sources, so the default build never sees it set
    csrc = os.path.join(ROOT, "qwen3-tts_amd", "csrc")
    text = "".join(open(os.path.join(csrc, f)).read() for f in os.listdir(csrc))
    for flags in m.VARIANTS.values():
        for f in flags:
            if not f.startswith("-DQTTS_"):        # (`pk`: code-generation flags, no macro)
                continue
            macro = f[2:].split("=")[0]
            assert re.search(r"#ifndef %s\s*\n#define %s 0" % (macro, macro), text), macro


def test_frame_step_kernels_issue_their_requests_back_to_back(libqtts):
    """Round 2 found the frame step's latency in the compiler's wait placement, not in the algorithm: a conditional load (or a
    load behind a run-time `contiguous ? arithmetic : page_table[...]`) makes the compiler wait for ALL outstanding loads before
    the next request (profiles/r02_isa_serialized_waits.md; attn_tk 9.2 -> 6.2 us once fixed).  This pins the fix in the gfx950
    code objects of the built library: the decode GEMM of batch <= 8 issues every request before its first `s_waitcnt vmcnt`,
    the two decode attentions issue their whole first burst (this step's row, norm weights, cache rows / speculative chunks)."""
    import shutil
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_waits
    ks = isa_waits.kernels(os.path.join(ROOT, "qwen3-tts_amd", "libqtts.so"))
    s8 = {n: ins for n, ins in ks.items() if "skinny8_kernel<" in n}
    assert len(s8) >= 30, len(s8)
    for n, ins in s8.items():
        n_loads = sum(isa_waits.is_load(i) for i in ins)
        assert n_loads >= 6, (n, n_loads)
        assert isa_waits.waits_inside_burst(ins, n_loads) == [], (n, "a wait between the requests of the launch")
    first_burst = {"attn_cp_kernel<unsigned short, true>": 25,       # 2 row + 3 norm / rope + 4 K + 16 V requests
                   "attn_tk_kernel<unsigned short, 2, true>": 29,    # 8 row + 5 norm / rope + 16 speculative K / V chunk requests
                   "attn_tk_kernel<unsigned short, 1, true>": 27}
    for key, n_first in first_burst.items():
        hit = [ins for n, ins in ks.items() if key in n]
        assert len(hit) == 1, key
        assert isa_waits.waits_inside_burst(hit[0], n_first) == [], (key, "a wait inside the first request burst")


def test_lds_dma_kernels_wait_for_their_requests_before_the_barrier(libqtts):
    """a workgroup barrier orders the waves, not the LDS-DMA data -- a tile staged with `global_load_lds` is in LDS once the
    REQUESTING wave's vmcnt has drained.  In the gfx950 code objects of the built library every `s_barrier` that follows a DMA request in
    program-text order has a `s_waitcnt vmcnt(0)` between the two (gemm_dma_kernel says so in its source; for the others this pins the
    compiler's fence placement)."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_waits
    d = isa_waits.dma_barriers(os.path.join(ROOT, "qwen3-tts_amd", "libqtts.so"))
    assert any("gemm_dma_kernel" in k for k in d) and any("resunit_kernel" in k for k in d), sorted(d)
    for k, (tot, bad) in d.items():
        assert tot >= 1 and bad == 0, (k, tot, bad)


def test_ring_tap_gemm_counted_waits_pinned_from_the_isa(libqtts):
    """Round 6: gemm_ring_kernel<NST, AH, RING_A> keeps the LDS-DMA requests of the tiles t + 2 .. t + NST - 1 in flight across the barrier of step t.
    From the gfx950 code objects: each instantiation has a prologue barrier, the steady-state steps' barriers and two draining steps'; the steady-state
    barriers are preceded by exactly `s_waitcnt vmcnt(LPS * (NST - 2))` (LPS = 2 requests per wave and step, 4 when the A tile rides in the ring);
    every barrier is preceded by `s_waitcnt lgkmcnt(0)` (a buffer is re-requested only when its last fragment reads are in registers)."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_waits
    d = isa_waits.ring_barriers(os.path.join(ROOT, "qwen3-tts_amd", "libqtts.so"))
    d = {k: v for k, v in d.items() if re.search(r"gemm_ring_kernel<\d+, \d+, (true|false), \d+, 0>", k)}      # (the product instantiations; ABL != 0 are measuring variants)
    assert len(d) == 8, sorted(d)                      # generic 4 / 6 / 8 deep, plain Linear 4 / 8, 7 taps unrolled 4 / 8, 2 taps unrolled 4
    for k, (nb, vm, lg, vmax) in d.items():
        m = re.search(r"gemm_ring_kernel<(\d+), (\d+), (true|false), (\d+), 0>", k)
        nst, lps = int(m.group(1)), 4 if m.group(3) == "true" else 2
        assert vm[-2:] == [[0], [0]], (k, vm[-3:])     # the two barriers of the split-K combine (behind the drained ring: a plain __syncthreads each)
        nb, vm, lg = nb - 2, vm[:-2], lg - 1 if lg == nb - 1 else lg - 2
        assert nb >= 7 and lg == nb, (k, nb, lg)       # prologue, the steady steps (plain and zeroing loops; whole slabs unrolled when the taps are known), the draining steps
        taps = int(m.group(4))
        if taps == 0:                                  # generic steady step: constant waits, then two draining steps behind the run-time switch
            assert all(v == [lps * (nst - 2)] for v in vm[1:-2]), (k, vm)
            assert all(v for v in vm[-2:]) and vmax <= 28, (k, vm, vmax)
        else:                                          # unrolled slabs: every wait an immediate; the steady slabs wait with the full count, the last slab drains to 0
            assert all(len(v) == 1 and v[0] <= lps * (nst - 2) for v in vm[1:]), (k, vm)
            full = sum(v == [lps * (nst - 2)] for v in vm[1:])
            assert full >= 2 * (2 * taps) and sum(v == [0] for v in vm[1:]) >= 2, (k, vm)     # (both loops: plain and zeroing)
            assert vmax <= 28, (k, vmax)


def test_granule_polling_loads_stay_inside_their_loops(libqtts):
    """Round 6, found on the MI355X (profiles/r06_skinny_ksplit.md): the first build of `skinny2_ks_kernel` polled its producers' granules ONCE --
    the sc1 buffer loads are read-only intrinsics, its polling loop held no store and no side effect, and the compiler hoisted the re-read
    out of the loop (the emulator runs workgroups in order and never needs a second read).  Pinned from the code objects: every kernel that
    polls granules (sc1 buffer loads) has such loads inside a loop (behind a backward branch)."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_waits
    d = isa_waits.polling_reloads(os.path.join(ROOT, "qwen3-tts_amd", "libqtts.so"))
    pollers = {k: v for k, v in d.items() if any(n in k for n in ("skinny2_ks_kernel", "cp_mlp_kernel", "cp_attn_o_kernel", "cp_layer_kernel"))}
    assert sum("skinny2_ks_kernel" in k for k in pollers) >= 4 and any("cp_mlp_kernel" in k for k in pollers), sorted(d)
    for k, (n, inside) in pollers.items():
        assert inside > 0, (k, n, "no sc1 load inside a loop: the polling loop was hoisted")


def test_build_toolchain_is_the_validated_one():
    """the build records the toolchain its flags and ISA-level assumptions were validated on and warns on another; here the image's hipcc IS
    that toolchain (a ROCm upgrade makes this test fail first: then the ISA pins of this file and the GPU contention test say whether the library holds)."""
    import importlib.util                              # (under its own module name: `build` is also the emulator builder's)
    spec = importlib.util.spec_from_file_location("qtts_build_tc", os.path.join(ROOT, "qwen3-tts_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    assert b.VALIDATED_TOOLCHAIN in b.toolchain_banner(hipcc), b.toolchain_banner(hipcc)[:300]


def test_product_library_has_no_packed_fp32_math(libqtts):
    """Round 5 (profiles/r05_packed_fp32_hazard.md): a `v_pk_mul_f32` / `v_pk_fma_f32` sequence returned wrong lanes 48-63 on the MI355X
    while another stream shared the device; the product library is built with packed fp32 math off.  Pinned from the code objects."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_waits
    n, names = isa_waits.packed_fp32(os.path.join(ROOT, "qwen3-tts_amd", "libqtts.so"))
    assert n == 0, (n, names[:5])


def test_fused_launches_fit_their_register_shares(libqtts):
    """The admission rule of the fused launches (talker_engine.hip: fused_admit) is an account of the register file: CP_SHARE = 184 registers
    per lane and SIMD for a workgroup of cp_attn_o_kernel / cp_mlp_kernel (4 waves, one per SIMD), CP_SHARE_F32 = 272 for the fp32
    instantiations of the two kernels (the exact parity mode: twice the operand registers).  Pinned from the code objects of the built library
    (`.vgpr_count` = arch + accumulator registers, allocated in granules of 8): every fused kernel within its share, LDS far from the limit, no
    scratch."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    ks = kernel_resources.kernels_of(libqtts)
    rows = [k for k in ks if any(n in k[".name"] for n in ("cp_attn_o_kernel", "cp_mlp_kernel"))]
    assert len(rows) >= 5, [k[".name"] for k in rows]
    n_f32 = 0
    for k in rows:
        regs = (k[".vgpr_count"] + 7) // 8 * 8
        waves_per_simd = k[".max_flat_workgroup_size"] // 256
        f32 = ("cp_mlp_kernelILb1E" in k[".name"]) or ("cp_attn_o_kernelILb" in k[".name"] and "ELb1EEE" in k[".name"])     # (<true, ...> / <.., .., true>)
        n_f32 += f32
        assert waves_per_simd == 1 and regs <= (272 if f32 else 184), (k[".name"], regs, waves_per_simd)
        assert 2 * k[".group_segment_fixed_size"] <= 160 * 1024 and k.get(".private_segment_fixed_size", 0) == 0, k[".name"]
    assert n_f32 >= 5
    src = open(os.path.join(ROOT, "qwen3-tts_amd", "csrc", "talker_engine.hip")).read()
    assert "CU_REG_BUDGET = 512, CP_SHARE = 184, CP_SHARE_F32 = 272;" in src
    # round 6: the layer launch (cp_layer.hip) -- both stages in the two launches' register share (bf16; 360 with the operators in registers in
    # fp32), and the account's second resource, LDS: dynamic (65 KB with the gate|up block at the released dims: two workgroups per compute unit
    # with an eighth of the LDS to spare, not three)
    rows = [k for k in ks if "cp_layer_kernel" in k[".name"]]
    assert len(rows) >= 6, [k[".name"] for k in rows]
    for k in rows:
        regs = (k[".vgpr_count"] + 7) // 8 * 8
        f32 = "cp_layer_kernelILb0ELb1E" in k[".name"] or "cp_layer_kernelILb1ELb1E" in k[".name"]      # (<QKV, F32 = true, ...>)
        assert k[".max_flat_workgroup_size"] == 256 and regs <= (360 if f32 else 192), (k[".name"], regs)
        assert k.get(".private_segment_fixed_size", 0) == 0, k[".name"]
    assert "CP_SHARE_LAYER = 192, CP_SHARE_LAYER_F32 = 360, CU_LDS_BUDGET = 160 * 1024" in src
    lay = open(os.path.join(ROOT, "qwen3-tts_amd", "csrc", "cp_layer.hip")).read()
    assert "static constexpr int GU = F32 ? 0 : KQ * 4 * 4 * 2 * ACT * 16;" in lay
    gu, att, mlp = 8 * 4 * 4 * 2 * 12 * 16, 4 * 1536 + 2 * 264 * 2 + 2 * 128 * 4 + (4 * 64 * 16 + 4 * 16 * 4), (4 * 64 * 2 * 16 + 4 * 16 * 4) + 4 * 2 * 64 * 16
    assert "static constexpr int TOTAL = GU + (ATT > MLP ? ATT : MLP);" in lay
    assert 2 * (gu + max(att, mlp)) <= 160 * 1024 * 7 // 8 < 3 * (gu + max(att, mlp))
    # round 6: the fused MLP launch at batch 9..32 (cp_mlp32.hip): 208 registers, two engines per device inside 7/8 of the register file
    rows = [k for k in ks if "cp_mlp32_kernel" in k[".name"]]
    assert len(rows) >= 2, [k[".name"] for k in rows]
    for k in rows:
        regs = (k[".vgpr_count"] + 7) // 8 * 8
        assert k[".max_flat_workgroup_size"] == 256 and regs <= 208 and k.get(".private_segment_fixed_size", 0) == 0, (k[".name"], regs)
        assert k[".group_segment_fixed_size"] <= 34 * 1024, k[".name"]
    assert "CP_SHARE_MLP32 = 208, CP_LDS_MLP32 = 34 * 1024;" in src and 2 * 208 <= 512 * 7 // 8


def test_option_table_through_the_c_abi(libqtts):
    """qtts_set_option / qtts_get_option (include/qtts.h, ABI v10): the A/B switches of the library are set through the C ABI -- the
    environment is looked at once per switch (the product path of QTTS_ENV: ADVICE r4), an override wins over it, removing the override
    falls back to it, names outside QTTS_* are refused."""
    import ctypes as C
    import subprocess
    code = r"""
import ctypes as C, os, sys
os.environ["QTTS_TEST_SWITCH_A"] = "env"
lib = C.CDLL(sys.argv[1])
lib.qtts_set_option.argtypes = [C.c_char_p, C.c_char_p]; lib.qtts_get_option.argtypes = [C.c_char_p, C.c_char_p, C.c_int32]
lib.qtts_last_error.restype = C.c_char_p
buf = C.create_string_buffer(64)
assert lib.qtts_get_option(b"QTTS_TEST_SWITCH_A", buf, 64) == 0 and buf.value == b"env"
assert lib.qtts_get_option(b"QTTS_TEST_SWITCH_B", buf, 64) == 1 and buf.value == b""
assert lib.qtts_set_option(b"QTTS_TEST_SWITCH_A", b"abi") == 0
assert lib.qtts_get_option(b"QTTS_TEST_SWITCH_A", buf, 64) == 0 and buf.value == b"abi"
assert lib.qtts_set_option(b"QTTS_TEST_SWITCH_A", None) == 0
assert lib.qtts_get_option(b"QTTS_TEST_SWITCH_A", buf, 64) == 0 and buf.value == b"env"
assert lib.qtts_set_option(b"PATH", b"x") != 0 and b"QTTS_" in lib.qtts_last_error()
print("ok")
"""
    r = subprocess.run([sys.executable, "-c", code, libqtts], capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if not k.startswith("QTTS_")})
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_ctypes_argtypes_match_the_header(libqtts):
    """Every entry point's ctypes signature in qwen3-tts_amd/_lib.py against its prototype in include/qtts.h: same number of
    parameters and the same class of each (pointer / int32 / int64 / float).  The emulator tests declare their own
    argtypes, so a slip here would otherwise first show on hardware."""
    import ctypes as C
    from qwen3_tts_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "qtts.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = dict(re.findall(r"\b(?:int|void|const char\*)\s+(qtts_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", hdr))
    lib = _lib.load_library()

    def klass_c(param):
        param = param.strip()
        if param in ("void", ""):
            return None
        if "*" in param:
            return "ptr"
        if re.search(r"\bint64_t\b|\blong long\b", param):
            return "i64"
        if re.search(r"\bfloat\b", param):
            return "f32"
        if re.search(r"\bdouble\b", param):
            return "f64"
        if re.search(r"\bint32_t\b|\bint\b|\buint32_t\b|\bunsigned\b", param):
            return "i32"
        raise AssertionError(f"unclassified parameter {param!r}")

    def klass_py(t):
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or getattr(t, "_type_", None) is not None and t.__name__.startswith("LP_"):
            return "ptr"
        return {C.c_int32: "i32", C.c_int: "i32", C.c_uint32: "i32", C.c_int64: "i64", C.c_uint64: "i64", C.c_float: "f32",
                C.c_double: "f64"}[t]

    checked = 0
    for name, params in protos.items():
        if name.startswith("qtts_debug_"):
            continue
        want = [k for k in (klass_c(p) for p in params.split(",")) if k]
        fn = getattr(lib, name)
        if fn.argtypes is None:
            assert not want, f"{name}: header has {len(want)} parameters, the binding declares none"
            continue
        got = [klass_py(t) for t in fn.argtypes]
        assert got == want, f"{name}: binding {got} vs header {want}"
        checked += 1
    assert checked >= 30, checked


def test_ctypes_struct_layouts_match_the_header(tmp_path):
    """The seven by-pointer structs of include/qtts.h against their ctypes mirrors: field names in order, offsets and total
    size as gcc lays them out (a probe program compiled from the header prints offsetof / sizeof)."""
    import ctypes as C
    from qwen3_tts_amd import _lib
    hdr_path = os.path.join(ROOT, "include", "qtts.h")
    hdr = re.sub(r"/\*.*?\*/", "", open(hdr_path).read(), flags=re.S)
    hdr = re.sub(r"//[^\n]*", "", hdr)
    structs = re.findall(r"typedef struct\s*\{(.*?)\}\s*(qtts_[a-z_]+)\s*;", hdr, flags=re.S)
    mirrors = {"qtts_codec_config": _lib.CodecConfigC, "qtts_talker_config": _lib.TalkerConfigC, "qtts_sampling": _lib.SamplingC,
               "qtts_encoder_config": _lib.EncoderConfigC, "qtts_speaker_config": _lib.SpeakerConfigC,
               "qtts_talker_stats": _lib.TalkerStatsC, "qtts_gemm_class": _lib.GemmClassC}
    assert {n for _, n in structs} == set(mirrors), {n for _, n in structs} ^ set(mirrors)
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{hdr_path}"', 'int main(void) {']
    fields = {}
    for body, name in structs:
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                ident = re.findall(r"([A-Za-z_][A-Za-z_0-9]*)\s*(?:\[[^\]]*\])?\s*$", part.strip())[0]
                names.append(ident)
        fields[name] = names
        lines.append(f'printf("{name} size %zu\\n", sizeof({name}));')
        lines += [f'printf("{name} {f} %zu\\n", offsetof({name}, {f}));' for f in names]
    lines += ['return 0; }']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    got = {}
    for ln in out:
        if ln.strip():
            n, f, v = ln.split()
            got.setdefault(n, {})[f] = int(v)
    for name, cls in mirrors.items():
        py_fields = [f[0] for f in cls._fields_]
        assert py_fields == fields[name], (name, py_fields, fields[name])
        assert C.sizeof(cls) == got[name]["size"], (name, C.sizeof(cls), got[name]["size"])
        for f in py_fields:
            assert getattr(cls, f).offset == got[name][f], (name, f)


def test_slaney_mel_filterbank_matches_librosa_published_values():
    """f4's audio front end (M:399-464 builds its filterbank with `librosa.filters.mel`, absent here): the Slaney scale and the
    filterbank restated in qwen3_tts_amd/speaker.py reproduce the numbers librosa's own documentation prints -- the examples of
    `librosa.hz_to_mel` (60 Hz -> 0.9; [110, 220, 440] -> [1.65, 3.3, 6.6]), `librosa.mel_to_hz` (3 -> 200; [1..5] -> 66.667 ...
    333.333), `librosa.mel_frequencies(n_mels=40)` (fmin = 0, fmax = 11025: the 40-entry table below, printed to 3 decimals) and
    `librosa.filters.mel(sr=22050, n_fft=2048)` (first row `[0., 0.016, ...]`) -- plus the closed-form anchors of the scale
    (1 kHz = 15 mel, 6.4 kHz = 42 mel) and the properties the "slaney" normalisation is defined by."""
    from qwen3_tts_amd.speaker import hz_to_mel_slaney, mel_filterbank_slaney, mel_frequencies_slaney, mel_to_hz_slaney
    assert abs(float(hz_to_mel_slaney(60)) - 0.9) < 1e-12
    assert np.allclose(hz_to_mel_slaney([110, 220, 440]), [1.65, 3.3, 6.6], atol=1e-12)
    assert abs(float(mel_to_hz_slaney(3)) - 200.0) < 1e-9
    assert np.allclose(mel_to_hz_slaney([1, 2, 3, 4, 5]), [66.667, 133.333, 200.0, 266.667, 333.333], atol=5e-4)
    assert abs(float(hz_to_mel_slaney(1000.0)) - 15.0) < 1e-12 and abs(float(hz_to_mel_slaney(6400.0)) - 42.0) < 1e-9
    published = [0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49, 1024.856,
                 1119.114, 1222.042, 1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393, 2470.47, 2697.686, 2945.799,
                 3216.731, 3512.582, 3835.643, 4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92, 7101.009, 7754.107, 8467.272,
                 9246.028, 10096.408, 11025.]
    assert np.abs(mel_frequencies_slaney(40, 0.0, 11025.0) - np.array(published)).max() < 6e-4          # (3 printed decimals)
    fb = mel_filterbank_slaney(22050, 2048, 128, 0.0, 11025.0)
    assert fb.shape == (128, 1025) and fb.dtype == np.float32
    assert abs(float(fb[0, 1]) - 0.016) < 5e-4 and float(fb[0, 0]) == 0.0 and float(fb[-1, -1]) == 0.0
    # "slaney" normalisation: every triangle has unit area in Hz (2 / width x peak 1 x width / 2), up to the FFT grid's sampling
    df = 22050.0 / 2048
    area = fb.astype(np.float64).sum(1) * df
    assert np.all(np.abs(area[8:] - 1.0) < 0.06), (area.min(), area.max())
    # each filter peaks between its neighbours' peaks, and the un-normalised triangles sum to 1 between the first and last centre
    mf = mel_frequencies_slaney(130, 0.0, 11025.0)
    tri = fb.astype(np.float64) / (2.0 / (mf[2:] - mf[:-2]))[:, None]
    grid = np.linspace(0.0, 11025.0, 1025)
    inner = (grid >= mf[1]) & (grid <= mf[-2])
    assert np.abs(tri.sum(0)[inner] - 1.0).max() < 1e-6
    # the configuration the speaker encoder uses (M:1944-1952): 24 kHz, n_fft 1024, 128 mels, fmin 0, fmax 12 kHz
    fb2 = mel_filterbank_slaney(24000, 1024, 128, 0.0, 12000.0)
    assert fb2.shape == (128, 513) and np.all(fb2 >= 0) and np.all(fb2.sum(1) > 0)


def test_from_pretrained_resolves_hub_ids_like_the_reference(monkeypatch, tmp_path):
    """IM:82-121 forwards its argument to `AutoModel.from_pretrained`, so the examples pass hub ids
    ("Qwen/Qwen3-TTS-12Hz-1.7B-CustomVoice/"): a local directory is used as it is, a hub id goes through
    `huggingface_hub.snapshot_download` (revision / cache_dir forwarded), and anything that cannot be resolved -- offline, unknown
    repo, a malformed name -- is the OSError HF raises."""
    import huggingface_hub
    from qwen3_tts_amd.model import resolve_checkpoint_dir
    d = tmp_path / "ckpt"
    d.mkdir()
    assert resolve_checkpoint_dir(str(d)) == str(d)
    calls = []

    def fake_snapshot(repo_id, **kw):
        calls.append((repo_id, kw))
        return str(d)
    monkeypatch.setattr(huggingface_hub, "snapshot_download", fake_snapshot)
    assert resolve_checkpoint_dir("Qwen/Qwen3-TTS-12Hz-1.7B-CustomVoice/", revision="main", device_map="cuda:0", dtype="x") == str(d)
    assert calls == [("Qwen/Qwen3-TTS-12Hz-1.7B-CustomVoice", {"revision": "main"})]       # only hub kwargs are forwarded

    def offline(repo_id, **kw):
        raise ConnectionError("no route to huggingface.co")
    monkeypatch.setattr(huggingface_hub, "snapshot_download", offline)
    with pytest.raises(OSError, match="could not be resolved as a hub id"):
        resolve_checkpoint_dir("Qwen/Qwen3-TTS-12Hz-0.6B-Base")
    with pytest.raises(OSError, match="neither a local directory nor a hub id"):
        resolve_checkpoint_dir(str(tmp_path / "a" / "b" / "missing"))
    # both loaders go through it (no engine is built: the error comes first)
    from qwen3_tts_amd.codec import Qwen3TTSTokenizer
    from qwen3_tts_amd.model import Qwen3TTSModel
    for cls in (Qwen3TTSModel, Qwen3TTSTokenizer):
        with pytest.raises(OSError):
            cls.from_pretrained("Qwen/Qwen3-TTS-Tokenizer-12Hz", device_map="cuda:0")


def test_audio_containers_other_than_wave_go_to_soundfile_when_present(monkeypatch):
    """IM:207-223 reads bytes with `soundfile.read(io.BytesIO(..), dtype="float32", always_2d=False)`: FLAC / OGG reach that call when
    the package is importable; without it (this image) a non-WAVE container is the ValueError it always was; RIFF/WAVE never leaves
    the module's own exact reader."""
    import io
    import types
    from qwen3_tts_amd import audio_io
    flac = b"fLaC" + bytes(64)
    monkeypatch.setitem(sys.modules, "soundfile", None)                  # import soundfile -> ImportError
    with pytest.raises(ValueError, match="RIFF/WAVE"):
        audio_io.read_audio_bytes(flac)
    seen = []
    fake = types.ModuleType("soundfile")

    def read(f, dtype=None, always_2d=None):
        seen.append((f.read(4), dtype, always_2d))
        return np.stack([np.linspace(-1, 1, 8), np.zeros(8)], 1).astype(np.float32), 16000
    fake.read = read
    monkeypatch.setitem(sys.modules, "soundfile", fake)
    a, sr = audio_io.read_audio_bytes(flac)
    assert seen == [(b"fLaC", "float32", False)] and sr == 16000 and a.shape == (8, 2) and a.dtype == np.float32
    # through the public path: base64 of a FLAC stream, down-mixed to mono like the reference (IM:218-219)
    import base64
    b64 = "data:audio/flac;base64," + base64.b64encode(flac).decode()
    mono, sr2 = audio_io.load_audio_to_np(b64)
    assert sr2 == 16000 and mono.shape == (8,) and np.allclose(mono, np.linspace(-1, 1, 8) / 2, atol=1e-6)
    # a WAVE file is still read by the exact in-module reader even when soundfile is there
    import struct
    pcm = np.array([0, 16384, -32768], "<i2").tobytes()
    wav = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 8000, 16000, 2, 16) + b"data" + struct.pack("<I", len(pcm)) + pcm
    n_before = len(seen)
    x, sr3 = audio_io.read_audio_bytes(wav)
    assert len(seen) == n_before and sr3 == 8000 and np.allclose(x, [0.0, 0.5, -1.0])


def test_pmc_traffic_tool_reduces_two_rocprofv3_passes_and_stamps_the_sources(tmp_path):
    """tools/pmc_traffic.py (what `bench.py`'s `roofline.traffic`, `frac_rocprof` and `fused_cp_launch` are read from): given a rocprofv3
    `--pmc FETCH_SIZE` database and a kernel-trace database it (1) reconstructs (N, K) of every `skinny8_kernel` dispatch from the
    instantiation and the grid, sums FETCH_SIZE x 1024 x 2 bytes against N x K x 2, (2) takes the launch-weighted duration of the same
    classes from the trace, (3) keeps the code predictor's fused launches apart (with / without the q|k|v front), and (4) stamps the JSON
    with the digest of csrc/skinny.hip + csrc/talker_engine.hip that `bench.py` compares before it uses the numbers."""
    import json, sqlite3, subprocess
    fetch_db, trace_db, out = str(tmp_path / "pmc.db"), str(tmp_path / "trace.db"), str(tmp_path / "pmc_traffic.json")
    sk = "void qtts::skinny8_kernel<1, 16, 4, true, 4>(void const*, float const*)"      # FS 16, NP 4, NW 4 -> K = 1024; 256 workgroups -> N = 4096
    fr = "void qtts::cp_attn_o_kernel<true, true>(qtts::CpAttnOParams)"
    ao = "void qtts::cp_attn_o_kernel<true, false>(qtts::CpAttnOParams)"
    con = sqlite3.connect(fetch_db)
    con.execute("create table counters_collection (kernel_name text, counter_name text, value real, grid_size_x int, workgroup_size_x int)")
    alg = 4096 * 1024 * 2
    rows = [(sk, "FETCH_SIZE", 1.05 * alg / 2048.0, 256 * 256, 256)] * 4 + [(sk, "OTHER", 1e9, 256 * 256, 256)]
    rows += [(fr, "FETCH_SIZE", 17.0e6 / 2048.0, 256 * 256, 256)] * 2 + [(ao, "FETCH_SIZE", 8.0e6 / 2048.0, 256 * 256, 256)]
    con.executemany("insert into counters_collection values (?, ?, ?, ?, ?)", rows); con.commit(); con.close()
    con = sqlite3.connect(trace_db)
    con.execute("create table kernels (name text, start int, end int, grid_size_x int, workgroup_size_x int)")
    con.executemany("insert into kernels values (?, ?, ?, ?, ?)",
                    [(sk, 0, 5000, 256 * 256, 256), (sk, 10000, 16000, 256 * 256, 256), (fr, 20000, 28000, 256 * 256, 256), (ao, 30000, 36000, 256 * 256, 256)])
    con.commit(); con.close()
    tool = os.path.join(ROOT, "tools", "pmc_traffic.py")
    r = subprocess.run([sys.executable, tool, "--fetch-db", fetch_db, "--trace-db", trace_db, "--source", "unit test", "--out", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.load(open(out))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    assert d["kernel_digest"] == pmc_traffic.kernel_digest() and len(d["kernel_digest"]) == 16
    rec = d["1.7b"]
    assert rec["dispatches"] == 4 and abs(rec["ratio_traffic_over_algorithmic"] - 1.05) < 1e-3 and rec["algorithmic_bytes_per_launch_skinny8"] == alg
    assert rec["rocprof_launches"] == 2 and abs(rec["rocprof_avg_launch_us"] - 5.5) < 1e-6
    assert abs(rec["frac_rocprof"] - alg / 5.5e-6 / 1e9 / 8000.0) < 1e-3
    assert rec["fused"]["front"] == {"dispatches": 2, "fetch_bytes_per_launch": 17000000, "rocprof_launches": 1, "rocprof_avg_launch_us": 8.0}
    assert rec["fused"]["attn_o"] == {"dispatches": 1, "fetch_bytes_per_launch": 8000000, "rocprof_launches": 1, "rocprof_avg_launch_us": 6.0}


def test_bench_fused_launch_report_accounts_for_the_bytes_that_left_the_decode_gemm():
    """`bench.fused_cp_report` (the `roofline.fused_cp_launch` object): at the 1.7B model's code-predictor dims, 14 passes x 4 layers take the
    launch with the q|k|v front (12.58 MB of operators each) and 14 x 1 the attention + o-projection launch (4.19 MB): 763 363 328 bytes per
    frame -- exactly what the GPU bench's `weight_bytes_per_frame_timed` is short of `weight_bytes_per_frame_model` (profiles/r04_bench_n1.json)."""
    import bench
    from qwen3_tts_amd.config import TalkerConfig
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    c = TalkerConfig.from_any(synth.cfg_dict(synth.talker_17b()))
    rec = {"front": {"rocprof_avg_launch_us": 8.475, "fetch_bytes_per_launch": 17972871}, "attn_o": {"rocprof_avg_launch_us": 6.229, "fetch_bytes_per_launch": 8135703},
           "_note": "ignored"}
    r = bench.fused_cp_report(c, rec)
    assert r["front"]["launches_per_frame"] == 56 and r["front"]["algorithmic_bytes_per_launch"] == 12582912
    assert r["attn_o"]["launches_per_frame"] == 14 and r["attn_o"]["algorithmic_bytes_per_launch"] == 4194304
    assert r["weight_bytes_per_frame"] == 763363328
    assert abs(r["front"]["frac_rocprof"] - 12582912 / 8.475e-6 / 1e9 / 8000.0) < 1e-3 and r["front"]["traffic"] == 17972871
    line = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")))["roofline"]
    assert line["weight_bytes_per_frame_model"] - line["weight_bytes_per_frame_timed"] == r["weight_bytes_per_frame"]
    assert bench.fused_cp_report(c, {}) == {}


def test_options_blocks_nest_and_restore_the_previous_override(libqtts):
    """`_lib.options` used to CLEAR its switches on exit, so a nested block dropped the outer block's override.  Now every
    switch goes back to the override it had before the block (or to none); the C table reads the environment once per name."""
    import subprocess, sys, textwrap
    code = textwrap.dedent(f"""
        import ctypes as C, os, sys
        sys.path.insert(0, {ROOT!r})
        os.environ["QTTS_TEST_SWITCH_E"] = "env0"
        from qwen3_tts_amd import _lib
        lib = _lib.load_library()
        def get(n):
            b = C.create_string_buffer(64)
            rc = lib.qtts_get_option(n.encode(), b, 64)
            return (rc, b.value.decode())
        with _lib.options(QTTS_TEST_SWITCH_N="outer"):
            assert get("QTTS_TEST_SWITCH_N") == (0, "outer")
            with _lib.options(QTTS_TEST_SWITCH_N="inner", QTTS_TEST_SWITCH_M="m"):
                assert get("QTTS_TEST_SWITCH_N") == (0, "inner") and get("QTTS_TEST_SWITCH_M") == (0, "m")
            assert get("QTTS_TEST_SWITCH_N") == (0, "outer"), "the inner block dropped the outer override"
            assert get("QTTS_TEST_SWITCH_M")[0] == 1
        assert get("QTTS_TEST_SWITCH_N")[0] == 1
        # the environment is looked at ONCE per name: a later change is not seen, an override still wins and falls back to the first value
        assert get("QTTS_TEST_SWITCH_E") == (0, "env0")
        os.environ["QTTS_TEST_SWITCH_E"] = "env1"
        assert get("QTTS_TEST_SWITCH_E") == (0, "env0")
        with _lib.options(QTTS_TEST_SWITCH_E="ovr"):
            assert get("QTTS_TEST_SWITCH_E") == (0, "ovr")
        assert get("QTTS_TEST_SWITCH_E") == (0, "env0")
        print("ok")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
