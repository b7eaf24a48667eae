"""CPU: the product's PYTHON layer -- engine classes, ctypes calls, tensor plumbing -- and the GPU parity TEST BODIES of the
rows that have not met the hardware yet, executed against the host-emulation build of the same C++ / HIP sources
(tests/hostemu: device memory = host memory, every kernel on the SIMT emulator).

tests/test_hostemu.py drives the C ABI with numpy arrays; what it cannot see is the Python between the user and that ABI
(`CodecEncoderEngine`, `SpeakerEncoderEngine`, `CodecDecoderEngine.stream_*`, `TalkerEngine.generate_stream`, the wrappers).
Here that Python runs unmodified: the library it loads is the emulation build (QTTS_LIBRARY), the one place that insists on
a HIP device (`_lib.hip_device`) is told to hand out the CPU device, and the handful of `torch.cuda.*` stream / context
calls are replaced by inert stand-ins.  The test functions called below are the ones in tests/test_gpu_parity.py, i.e. exactly
what `QTTS_EXPERIMENTAL=1 pytest -m gpu` will run on the MI355X -- so a slip in the Python glue, or in the tests themselves,
shows here first.  Nothing of this is reachable from the product: the product refuses a CPU device."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def glue():
    sys.path.insert(0, os.path.join(HERE, "hostemu"))
    import pyshim
    pyshim.install()
    try:
        import test_gpu_parity as gp
        yield gp
    finally:
        pyshim.uninstall()


def _codec_tiny(gp, golden_dir):
    import synth
    from qwen3_tts_amd.codec import CodecDecoderEngine
    c = synth.codec_tiny()
    w = gp._td(synth.codec_weights(c))
    g = np.load(os.path.join(golden_dir, "codec_tiny.npz"))
    return c, w, g, CodecDecoderEngine(c, w, compute_dtype=torch.float32, device="cpu", max_batch=4, max_frames=64)


def _talker_tiny(gp, golden_dir):
    import synth
    t = synth.talker_tiny()
    return t, gp._td(synth.talker_weights(t)), np.load(os.path.join(golden_dir, "talker_tiny.npz"))


FULL = os.environ.get("QTTS_GLUE_FULL") == "1"      # the default CPU suite keeps to the cases that add something new per minute


@pytest.mark.skipif(not FULL, reason="QTTS_GLUE_FULL=1 (green on the MI355X in round 2: tests/test_gpu_parity.py runs the same body through the HIP build)")
def test_state_carrying_codec_stream_python_path(glue, golden_dir):
    """`CodecDecoderEngine.stream_begin / stream_push`: the gated GPU test body, on the emulator."""
    glue.test_codec_incremental_stream_equals_forward(_codec_tiny(glue, golden_dir))


def test_codec_encoder_python_path(glue, golden_dir):
    """`CodecEncoderEngine.encode_padded / encode` (f3) against the reference's encoder golden: the gated GPU test body."""
    glue.test_codec_encoder_codes_vs_reference_golden("cpu", golden_dir)


def test_speaker_encoder_python_path(glue):
    """`SpeakerEncoderEngine.embed / extract_speaker_embedding` (f4) against the oracle: the gated GPU test body."""
    glue.test_speaker_embedding_vs_oracle("cpu")




@pytest.mark.skipif(not FULL, reason="QTTS_GLUE_FULL=1 (the streaming wrapper test below goes through generate_stream too)")
@pytest.mark.parametrize("graph", [False, True])
def test_talker_generate_stream_python_path(glue, golden_dir, graph):
    """`TalkerEngine.generate_stream` (streaming output): the gated GPU test body, eager and through the captured frame graph."""
    glue.test_talker_generate_stream_equals_generate(_talker_tiny(glue, golden_dir), "cpu", graph)


@pytest.mark.skipif(not FULL, reason="QTTS_GLUE_FULL=1")
def test_validated_python_paths_still_hold_on_the_emulator(glue, golden_dir):
    """Two of the hardware-validated test bodies through the same harness, as its own control: if these fail here the harness
    is wrong, not the product."""
    glue.test_codec_stream_decoder_packets(_codec_tiny(glue, golden_dir))
    glue.test_talker_tiny_greedy_bit_exact(_talker_tiny(glue, golden_dir), "cpu", False)


@pytest.mark.skipif(not FULL, reason="QTTS_GLUE_FULL=1 (green on the MI355X in round 2: tests/test_gpu_parity.py runs the same body through the HIP build)")
def test_voice_clone_wrapper_end_to_end_python_path(glue, tmp_path):
    """`create_voice_clone_prompt` + `generate_voice_clone` from a reference WAVE file (BASELINE config 5's call sequence):
    audio_io -> codec encoder -> speaker encoder -> device prompt assembly -> talker -> decoder -> ICL cut, every engine the
    emulated build of its real sources -- the gated GPU test body."""
    glue.test_wrapper_voice_clone_from_waveform_end_to_end("cpu", tmp_path)


@pytest.mark.skipif(not FULL, reason="QTTS_GLUE_FULL=1 (green on the MI355X in round 2: tests/test_gpu_parity.py runs the same body through the HIP build)")
def test_stream_custom_voice_wrapper_python_path(glue):
    """`Qwen3TTSModel.stream_custom_voice` (PCM packets) equals `generate_custom_voice`: the gated GPU test body."""
    glue.test_wrapper_stream_custom_voice_equals_one_shot("cpu")


def test_round1_advice_fixes_python_path(glue, golden_dir):
    """Out-of-range codec codes raise IndexError (device flag / host check), and the default Philox seed advances between
    calls while `torch.manual_seed` reproduces a run: the GPU test bodies, on the emulator."""
    glue.test_codec_edge_cases(_codec_tiny(glue, golden_dir))
    glue.test_default_seed_advances_and_manual_seed_reproduces(_talker_tiny(glue, golden_dir), "cpu")


def test_teacher_forcing_python_path(glue, golden_dir):
    """`TalkerEngine.generate(teacher_codes=...)` / `qtts_talker_set_teacher`: the GPU test body on the emulator."""
    glue.test_teacher_forcing_tiny_reproduces_golden(_talker_tiny(glue, golden_dir), "cpu")


def test_top_p_zero_and_integer_like_top_k_python_path(glue, golden_dir):
    """HF's TopPLogitsWarper accepts `top_p == 0` (the nucleus cut removes everything, `min_tokens_to_keep = 1` puts the
    top token back): the sampled token is the greedy one whatever the seed; only `top_p < 0` or `> 1` raises; numpy integers are
    accepted for `top_k`."""
    from qwen3_tts_amd.talker import TalkerEngine, _resolve_top
    assert _resolve_top(50, 0.0) == (1, 1.0) and _resolve_top(np.int64(7), 0.5) == (7, 0.5) and _resolve_top(None, None) == (0, 1.0)
    t, w, g = _talker_tiny(glue, golden_dir)
    eng = TalkerEngine(t, w, weight_dtype=torch.float32, device="cpu", max_batch=4, max_seq=64, use_graph=False)
    args = [torch.from_numpy(g[k]) for k in ("embeds", "mask", "trailing", "tts_pad")]
    sup = glue._suppress(t)
    greedy = eng.generate(*args, max_new_tokens=1, do_sample=False, suppress_tokens=sup).tokens[:, 0].numpy()
    for seed in (1, 2):
        o = eng.generate(*args, max_new_tokens=1, do_sample=True, top_k=np.int64(0), top_p=0.0, temperature=0.8, suppress_tokens=sup, seed=seed)
        assert np.array_equal(o.tokens[:, 0].numpy(), greedy)
    for bad in (dict(top_p=1.5), dict(top_p=-0.1), dict(top_k=2.5), dict(subtalker_top_k=-3)):
        with pytest.raises(ValueError):
            eng.generate(*args, max_new_tokens=1, do_sample=True, suppress_tokens=sup, **bad)


def test_shared_device_engines_keep_the_decode_gemms_python_path(glue):
    """Round 6: `TalkerEngine(shared_device=True)` -- an engine that will run beside another talker engine on its device (clone-shard: two engines
    per GPU) keeps the decode GEMMs where an engine alone on the device runs the code predictor's MLP as one launch at batch 9..32
    (csrc/cp_mlp32.hip; profiles/r06_cp_mlp32.md: 86.6 k vs 130 k tokens/s for the two-engine job).  Through the Python class on the emulator:
    `cp_mlp_per_step` says which path each engine took, the flag leaves no override behind in the option table, and both engines decode the
    same batch of 10 to the same shapes (their bf16 sums differ in order only)."""
    import dataclasses
    import synth
    from qwen3_tts_amd import _lib
    from qwen3_tts_amd.talker import TalkerEngine
    t = dataclasses.replace(synth.talker_tiny(), num_code_groups=3, cp_hidden_size=256, cp_intermediate_size=1024, cp_num_hidden_layers=2,
                            cp_num_attention_heads=16, cp_num_key_value_heads=8, cp_head_dim=128)
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(65), t, [3 + i % 3 for i in range(10)], 2, scale=0.5)
    sup = glue._suppress(t)
    outs = {}
    for shared in (False, True):
        eng = TalkerEngine(t, w, weight_dtype=torch.bfloat16, device="cpu", max_batch=16, max_seq=64, use_graph=True, shared_device=shared)
        assert _lib.get_override("QTTS_CP_MLP32") is None
        o = eng.generate(emb, mask, tr, pad, max_new_tokens=3, min_new_tokens=3, do_sample=False, subtalker_dosample=False, suppress_tokens=sup)
        st = eng.stats()
        assert st["cp_mlp_per_step"] == (0 if shared else (t.num_code_groups - 2) * t.cp_num_hidden_layers), (shared, st)
        assert st["cp_fused_giveups"] == 0
        outs[shared] = o.codes.numpy()
        del eng
    assert outs[False].shape == outs[True].shape == (10, 2, 3)
    assert float((outs[False] == outs[True]).mean()) >= 0.9


def test_codec_graph_replay_python_path(glue):
    """`CodecDecoderEngine.decode_padded` called repeatedly on the same buffers + `stats()`: the GPU test body on the emulator (whether a
    call hits the graph cache depends on the allocator handing the output block back; both outcomes are checked for equal results)."""
    glue.test_codec_decode_calls_replay_as_graphs_on_the_gpu("cpu")


def test_sampled_path_body_on_the_emulator(glue):
    """The GPU suite's sampled-path test body (`_sampled_path_body`: the engine's own raw logits -> the oracle's HF processors ->
    rank-bucket chi-square + the per-draw check that u = Philox(seed; step, row, stream) falls into the token's inverse-CDF interval in
    the kernel's candidate order) on the emulator: a bf16 engine through the captured frame graph with both fused launches, a talker
    vocabulary of 2304 (`sample_kernel_v2<12>`) and a code-predictor vocabulary of 256 (`<8>`), a few seeds.  What it pins here is the
    body itself -- the Philox restatement, the counter layout (the code predictor's samplers read the same step counter as the talker's),
    the slot order -- so that the 2000-seed run on the MI355X starts from a checked test; the draw-by-draw check is exact at any seed count."""
    import dataclasses
    import synth
    t = dataclasses.replace(synth.talker_tiny(), vocab_size=2304, num_code_groups=4, cp_hidden_size=256, cp_intermediate_size=1024,
                            cp_num_hidden_layers=2, cp_num_attention_heads=16, cp_num_key_value_heads=8, cp_head_dim=128)
    w = glue._td(synth.talker_weights(t, with_text=False))
    fused = t.cp_num_hidden_layers * (t.num_code_groups - 2)
    ev = glue._sampled_path_body("cpu", t, w, torch.bfloat16, True, 5, 3, 3, fused)
    assert ev["draws"] >= 5 * 3 * t.num_code_groups - 12 and ev["worst_u_gap"] <= 2e-5
    ev1 = glue._sampled_path_body("cpu", t, w, torch.bfloat16, True, 4, 3, 3, fused, rep=1.5)
    assert ev1["worst_u_gap"] <= 2e-5
    ev0 = glue._sampled_path_body("cpu", t, w, torch.float32, False, 4, 1, 3, None)
    assert ev0["draws"] == 12


def test_speaker_embed_many_buckets_by_length_python_path(glue):
    """Round 6 (`create_voice_clone_prompt`, IM:440-455: the reference embeds clip by clip): `SpeakerEncoderEngine.embed_many` runs clips of
    equal length as one batch of up to max_batch rows and buckets ragged clips by exact length -- in the callers' order, every row equal to
    the same clip embedded on its own (a row's x-vector does not depend on its neighbours), and equal to the oracle."""
    import synth
    import speaker_ref
    from qwen3_tts_amd.speaker import SpeakerEncoderEngine
    c = synth.speaker_small()
    w = synth.speaker_weights(c)
    n = 4096
    eng = SpeakerEncoderEngine(synth.cfg_dict(c), glue._td(w), compute_dtype=torch.float32, device="cpu", max_batch=3, max_samples=n)
    a = synth.rand_audio(7, 6, n)
    clips = [a[0], a[1][:3000], a[2], a[3], a[4][:3000], a[5]]          # lengths 4096 x 4 (two calls: 3 + 1) and 3000 x 2
    many = [e.numpy() for e in eng.embed_many(clips)]
    assert len(many) == 6
    for i, clip in enumerate(clips):
        solo = eng.embed(torch.from_numpy(clip[None])).numpy()[0]
        assert np.abs(solo - many[i]).max() <= 1e-5, i
    with torch.no_grad():
        ref = speaker_ref.extract_speaker_embedding(glue._td(w), c, clips[1], 24000).numpy()
    assert np.abs(ref - many[1]).max() <= 2e-4 * max(1.0, float(np.abs(ref).max()))
