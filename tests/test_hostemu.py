"""The product's REAL C++ and HIP sources on the CPU.

tests/hostemu builds csrc/*_engine.hip as host C++ (device memory = host memory) and EVERY kernel source of csrc/ for a
small SIMT emulator (hostemu/simt.h: one fiber per thread, workgroup barriers, wave collectives -- shuffles, DPP row
rotations, ballots, MFMA 16x16x32 bf16 / 16x16x4 f32, LDS-DMA).  What runs here is therefore the product's own code end
to end: finalize() weight repacking, buffer rotation, strides, streaming carries, the C ABI, and the kernels' tiling,
LDS traffic and cross-lane reductions -- checked against the oracle and the reference goldens.  One stand-in remains, for
speed only: the large codec-decoder tests route the tap GEMM through plain loops (hostemu/cpu_gemm_tap.cpp); the
encoder, the speaker encoder, the short decoder / stream cases and the kernel-level cases below always run the real
gemm_tap.hip, and QTTS_HOSTEMU_FULL=1 runs the whole file with it (17 min; result recorded in DESIGN.md).
It is how the code paths that have not had a hardware run yet (state-carrying stream decode, codec encoder, speaker
encoder, resumable generation) are exercised in round 1, and it keeps the validated paths under a CPU regression test.
It says nothing about performance, and it encodes the MFMA / DPP lane layouts as this project uses them (validated on
hardware by the GPU suite)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

import codec_ref
import codec_enc_ref
import synth
from qwen3_tts_amd import _lib
from qwen3_tts_amd.config import CodecDecoderConfig, CodecEncoderConfig

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    sys.path.insert(0, os.path.join(HERE, "hostemu"))
    import build as hostemu_build
    lib = C.CDLL(hostemu_build.build())
    vp, i32, i64p = C.c_void_p, C.c_int32, C.POINTER(C.c_int64)
    lib.qtts_last_error.restype = C.c_char_p
    lib.qtts_codec_create.argtypes = [C.POINTER(_lib.CodecConfigC), C.POINTER(vp)]
    lib.qtts_codec_destroy.argtypes = [vp]; lib.qtts_codec_destroy.restype = None
    lib.qtts_codec_bind.argtypes = [vp, C.c_char_p, vp, i32, i32, i64p]
    lib.qtts_codec_finalize.argtypes = [vp]
    lib.qtts_codec_forward.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    lib.qtts_codec_decode.argtypes = [vp, vp, i32, i32, i32, i32, vp, i64p, vp]
    lib.qtts_codec_stream_begin.argtypes = [vp, i32]
    lib.qtts_codec_stream_push.argtypes = [vp, vp, i32, vp, vp]
    lib.qtts_encoder_create.argtypes = [C.POINTER(_lib.EncoderConfigC), C.POINTER(vp)]
    lib.qtts_encoder_destroy.argtypes = [vp]; lib.qtts_encoder_destroy.restype = None
    lib.qtts_encoder_bind.argtypes = [vp, C.c_char_p, vp, i32, i32, i64p]
    lib.qtts_encoder_finalize.argtypes = [vp]
    lib.qtts_encoder_frames.argtypes = [vp, C.c_int64, i64p]
    lib.qtts_encoder_encode.argtypes = [vp, vp, i32, i32, vp, vp]
    fp = C.POINTER(C.c_float)
    lib.hostemu_set_real_gemm.argtypes = [i32]; lib.hostemu_set_real_gemm.restype = None
    lib.hostemu_set_fiber_order.argtypes = [i32]; lib.hostemu_set_fiber_order.restype = None
    lib.hostemu_set_block_order.argtypes = [i32]; lib.hostemu_set_block_order.restype = None
    lib.hostemu_cp_layer_front.argtypes = [vp, i32, vp, vp, C.c_float, vp, vp, C.c_float, vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, C.c_uint32, i32]
    lib.hostemu_cp_attn_o.argtypes = [vp, i32, i32, vp, vp, C.c_float, vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, vp, i32, C.c_uint32]
    lib.hostemu_cp_mlp.argtypes = [vp, i32, vp, vp, vp, C.c_float, vp, i32, i32, vp, vp, vp, i32, C.c_uint32, i32]
    lib.hostemu_cp_mlp32.argtypes = [vp, i32, vp, vp, vp, C.c_float, vp, i32, i32, vp, vp, vp, C.c_uint32]
    lib.hostemu_gemm_tap.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, C.POINTER(i32), vp, vp, vp, i32, vp, vp, i32, vp, i32, i32]
    lib.hostemu_skinny.argtypes = [vp, i32, i32, vp, i32, i32, vp, i32, C.c_float, vp, vp, i32, i32, vp, i32, i32]
    lib.hostemu_gemm_tap16.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp, i32, vp, vp, vp]
    lib.hostemu_skinny_bf16x.argtypes = [vp, i32, i32, vp, i32, i32, vp, i32, C.c_float, vp, vp, i32, i32, vp, i32, i32, vp]
    lib.hostemu_resunit.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.hostemu_sample.argtypes = [vp, i32, i32, i32, vp, i32, i32, C.c_float, i32, i32, vp, i32, i32, C.c_float, C.c_float,
                                   C.c_uint64, C.c_uint32, i32, vp]
    lib.hostemu_attn_decode.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, C.c_float, vp, vp, i32, vp, vp, vp, i32, i32, vp,
                                        i32, i32]
    return lib


FULL = os.environ.get("QTTS_HOSTEMU_FULL") == "1"


class real_gemm:
    """Route the engines' tap GEMMs through the real gemm_tap.hip kernels inside the block."""

    def __init__(self, lib):
        self.lib = lib

    def __enter__(self):
        self.lib.hostemu_set_real_gemm(1)

    def __exit__(self, *a):
        self.lib.hostemu_set_real_gemm(1 if FULL else 0)


def _bf16_round(x):
    """fp32 -> nearest-even bf16, returned as (fp32 values, uint16 bits)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    return (r << 16).astype(np.uint32).view(np.float32).reshape(x.shape), r.astype(np.uint16).reshape(x.shape)


ACT_NONE, ACT_GELU, ACT_SWIGLU, ACT_SNAKE, ACT_SILU, ACT_SWIGLU8 = 0, 1, 2, 3, 4, 5     # csrc/kernels.h enum Act


def _gemm_tap_ref(A, T, W, shift, bias, scale, res, ea, ib, act):
    M, (taps, N, K) = A.shape[0], W.shape
    acc = np.zeros((M, N), np.float64)
    for tap in range(taps):
        for m in range(M):
            if (m % T) + shift[tap] >= 0:
                acc[m] += W[tap].astype(np.float64) @ A[m + shift[tap], :K].astype(np.float64)
    if act == ACT_SWIGLU:
        g = acc.reshape(M, N // 32, 2, 16)
        return ((g[:, :, 0] / (1 + np.exp(-g[:, :, 0]))) * g[:, :, 1]).reshape(M, N // 2)
    v = acc + (bias if bias is not None else 0)
    if act == ACT_GELU:
        from scipy.special import erf
        v = 0.5 * v * (1 + erf(v / np.sqrt(2)))
    elif act == ACT_SNAKE:
        v = v + ib * np.sin(v * ea) ** 2
    elif act == ACT_SILU:
        v = v / (1 + np.exp(-v))
    if scale is not None:
        v = v * scale
    if res is not None:
        v = v + res
    return v


@pytest.mark.parametrize("bf16", [0, 1])
def test_gemm_tap_kernel_real_source(emu, bf16):
    """gemm_tap.hip itself (LDS tiles, MFMA, per-tap row shifts with sequence-start zeroing, every epilogue, the wide-K
    variant) on the emulator against float64 numpy, for the shapes the engines use: ragged M and N, strided lda / ldc."""
    g = np.random.default_rng(40 + bf16)
    cases = [  # M, T, N, K, taps(shifts), act, lda_pad, with bias/scale/res
        (37, 37, 48, 32, [0], ACT_NONE, 0, (1, 0, 0)),
        (40, 20, 80, 64, [-2, -1, 0], ACT_GELU, 4, (1, 1, 1)),
        (33, 11, 64, 96, [-6, -3, 0], ACT_SNAKE, 0, (1, 0, 1)),
        (18, 9, 64, 32, [-1, 0], ACT_SILU, 8, (1, 0, 0)),
        (50, 25, 128, 64, [0], ACT_SWIGLU, 0, (0, 0, 0)),
        (70, 70, 16, 160, [-7, -5, -4, -3, -2, -1, 0], ACT_NONE, 0, (0, 1, 0)),
        (24, 24, 128, 512, [0], ACT_NONE, 0, (1, 0, 1)),          # wide-K variant (bf16: K % 128 == 0, K >= 512)
    ]
    for (M, T, N, K, shift, act, pad, (hb, hs, hr)) in cases:
        lda = K + pad
        A = (g.standard_normal((M, lda)) * 0.5).astype(np.float32)
        W = (g.standard_normal((len(shift), N, K)) / np.sqrt(K * len(shift))).astype(np.float32)
        bias = g.standard_normal(N).astype(np.float32) if hb else None
        scale = g.standard_normal(N).astype(np.float32) if hs else None
        No = N // 2 if act == ACT_SWIGLU else N
        res = g.standard_normal((M, No + 4)).astype(np.float32) if hr else None
        ea = np.exp(g.standard_normal(N) * 0.3).astype(np.float32)
        ib = (1 / (np.exp(g.standard_normal(N) * 0.3) + 1e-9)).astype(np.float32)
        if bf16:
            Wv, Wbits = _bf16_round(W)
            Av = _bf16_round(A)[0]
            Wdev = Wbits
        else:
            Wv, Av, Wdev = W, A, W
        want = _gemm_tap_ref(Av, T, Wv, shift, bias, scale, res[:, :No] if hr else None, ea, ib, act)
        ldc = No + 8
        out = np.full((M, ldc), 7.0, np.float32)
        sh = (C.c_int32 * len(shift))(*shift)
        rc = emu.hostemu_gemm_tap(_ptr(A), lda, M, T, _ptr(Wdev), N, K, len(shift), sh, _ptr(bias) if hb else None,
                                  _ptr(scale) if hs else None, _ptr(res) if hr else None, No + 4, _ptr(ea), _ptr(ib), act,
                                  _ptr(out), ldc, bf16)
        assert rc == 0, ((M, N, K, shift, act), (emu.qtts_last_error() or b"").decode())
        tol = (2e-3 if bf16 else 2e-5) * max(1.0, float(np.abs(want).max()))
        assert np.abs(out[:, :No] - want).max() <= tol, (M, N, K, shift, act, float(np.abs(out[:, :No] - want).max()))
        assert np.all(out[:, No:] == 7.0), "wrote outside its columns"


_WIDE_TILE_REF = {}


@pytest.mark.parametrize("tile", [128128128, 128064128, 64128128, 64064128, 64064256, 64128256, 32064256, 32032256])
def test_gemm_wide_kernel_every_tile_real_source(emu, tile):
    """The small-grid GEMM of the talker prefill and the codec transformer (gemm_wide_kernel, round 3: 64-row tiles, 256-wide k-steps,
    bf16 activations): every tile instantiation, forced through the test hook, with fp32 and with bf16 activations, ragged M, a
    residual / bias epilogue and the SwiGLU epilogue, against float64 numpy.  (A tile the shape or the registers do not admit -- the
    256-wide step with 128 fp32 columns -- runs as 128 x 128, which the first parameter covers.)  Round 6: the 32-row tiles (launches of at most
    128 rows; 32 x 32 has no SwiGLU epilogue) -- the third shape is theirs, and every tile must agree with 128 x 128 on it BIT FOR BIT (the MFMA
    sequence per output element does not depend on the tile: what lets the chooser change tiles under the goldens)."""
    g = np.random.default_rng(tile % 1000 + tile // 1000000)
    g3 = np.random.default_rng(3)
    emu.qtts_debug_gemm_wide_tile.argtypes = [C.c_int32]; emu.qtts_debug_gemm_wide_tile.restype = None
    emu.qtts_debug_gemm_wide_tile(tile)
    try:
        for (M, N, K, act, hb, hr) in ((150, 192, 512, ACT_NONE, 1, 1), (70, 128, 1024, ACT_SWIGLU, 0, 0), (101, 192, 768, ACT_NONE, 1, 1)):
            if M == 101:
                g = g3                                                   # the same operands for every tile
            A = (g.standard_normal((M, K)) * 0.5).astype(np.float32)
            W = (g.standard_normal((1, N, K)) / np.sqrt(K)).astype(np.float32)
            bias = g.standard_normal(N).astype(np.float32) if hb else None
            No = N // 2 if act == ACT_SWIGLU else N
            res = g.standard_normal((M, No + 4)).astype(np.float32) if hr else None
            Wv, Wbits = _bf16_round(W)
            Av, Abits = _bf16_round(A)
            want = _gemm_tap_ref(Av, M, Wv, [0], bias, None, res[:, :No] if hr else None, None, None, act)
            ldc = No + 8
            sh = (C.c_int32 * 1)(0)
            tol = 2e-3 * max(1.0, float(np.abs(want).max()))
            out = np.full((M, ldc), 7.0, np.float32)
            rc = emu.hostemu_gemm_tap(_ptr(A), K, M, M, _ptr(Wbits), N, K, 1, sh, _ptr(bias) if hb else None, None,
                                      _ptr(res) if hr else None, No + 4, None, None, act, _ptr(out), ldc, 1)
            assert rc == 0, (emu.qtts_last_error() or b"").decode()
            assert np.abs(out[:, :No] - want).max() <= tol and np.all(out[:, No:] == 7.0), (tile, M, N, K, "fp32 activations")
            out16 = np.full((M, ldc), 7.0, np.float32)
            rc = emu.hostemu_gemm_tap16(_ptr(Abits), K, M, M, _ptr(Wbits), N, K, 1, sh, _ptr(bias) if hb else None,
                                        _ptr(res) if hr else None, No + 4, None, None, act, _ptr(out16), ldc, None, None, None)
            assert rc == 0, (emu.qtts_last_error() or b"").decode()
            assert np.abs(out16[:, :No] - want).max() <= tol and np.all(out16[:, No:] == 7.0), (tile, M, N, K, "bf16 activations")
            if tile == 128128128:                                   # the same rounding either way: bf16 activations change nothing
                assert np.array_equal(out16[:, :No], out[:, :No])
            if M == 101:
                ref = _WIDE_TILE_REF.setdefault("out", out[:, :No].copy())
                assert np.array_equal(out[:, :No], ref) and np.array_equal(out16[:, :No], ref), (tile, "differs from the first tile's bits")
    finally:
        emu.qtts_debug_gemm_wide_tile(-1)


@pytest.mark.parametrize("C_, T, skip", [(96, 300, 0), (96, 517, 40), (32, 70, 0)])
def test_final_conv_kernels_real_source(emu, C_, T, skip):
    """The codec's last layer (V2:884: causal Conv1d(C -> 1, k = 7) + clamp), both kernels: `final_conv_kernel` on the fp32 tensor and
    round 3's `final_conv16_kernel` on the bf16 copy the last residual unit leaves -- two sequences (the causal left padding must not
    read the previous sequence's tail), T not a multiple of the 64 / 256 outputs per workgroup, the first `skip` samples dropped (chunked
    decode), pre-clamp output returned, values beyond +-1 present so that the clamp is exercised -- against float64 numpy."""
    g = np.random.default_rng(C_ + T)
    B = 2
    x = (g.standard_normal((B, T, C_)) * 0.8).astype(np.float32)
    w = (g.standard_normal((7, C_)) / np.sqrt(C_)).astype(np.float32)
    bias = 0.05
    xb, xbits = _bf16_round(x)
    emu.hostemu_final_conv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p] + [C.c_int] * 5
    for name, xin, tol in (("fp32", x, 2e-5), ("bf16", xb, 2e-5)):
        want = np.zeros((B, T), np.float64)
        xp = np.concatenate([np.zeros((B, 6, C_)), xin.astype(np.float64)], axis=1)
        for k in range(7):
            want += (xp[:, k:k + T] * w[k].astype(np.float64)).sum(-1)
        want += bias
        stride = T - skip + 3
        wav = np.full((B, stride), 7.0, np.float32)
        pre = np.full((B, stride), 7.0, np.float32)
        rc = emu.hostemu_final_conv(_ptr(x) if name == "fp32" else None, _ptr(xbits) if name == "bf16" else None, _ptr(w), bias,
                                    _ptr(wav), _ptr(pre), B, T, C_, stride, skip)
        assert rc == 0, (emu.qtts_last_error() or b"").decode()
        n = T - skip
        assert np.abs(pre[:, :n] - want[:, skip:]).max() <= tol * max(1.0, float(np.abs(want).max())), name
        assert np.array_equal(wav[:, :n], np.clip(pre[:, :n], -1.0, 1.0)) and float(np.abs(pre[:, :n]).max()) > 1.0, name
        assert np.all(wav[:, n:] == 7.0) and np.all(pre[:, n:] == 7.0), "wrote outside its samples"


def test_gemm_wide_tile_chooser_follows_its_three_bounds(emu):
    """`wide_tile()` (gemm_tap.hip) picks the small-grid GEMM's tile from three measured bounds (profiles/r03_gemm_small_tiles.md).  Host
    code only.  Pinned here: every choice is an instantiation that exists for the shape; where 128-row tiles would leave most CUs idle
    (the prefill's q|k|v, o and down projections at 512 rows, the codec transformer's at 125 / 1000 rows) it at least doubles the workgroups; where every
    row tile streams a 50 MB operator (gate|up) or the grid already fills the chip (2048 rows) it keeps 128 x 128; a 256-wide k-step
    only with 64 rows, and with fp32 activations only on 64 columns (registers)."""
    f = emu.qtts_debug_gemm_wide_choice
    f.argtypes = [C.c_int32] * 4 + [C.POINTER(C.c_int32)] * 3; f.restype = None
    emu.qtts_debug_gemm_wide_tile.argtypes = [C.c_int32]; emu.qtts_debug_gemm_wide_tile.restype = None
    emu.qtts_debug_gemm_wide_tile(0)                                     # the chooser itself, whatever QTTS_GEMM_WIDE_TILE says
    def pick(M, N, K, a16):
        bm, bn, bk = C.c_int32(), C.c_int32(), C.c_int32()
        f(M, N, K, a16, C.byref(bm), C.byref(bn), C.byref(bk))
        return bm.value, bn.value, bk.value
    try:
        for a16 in (0, 1):
            for (M, N, K) in ((512, 4096, 2048), (512, 2048, 2048), (512, 12288, 2048), (512, 2048, 6144), (2048, 4096, 2048), (2048, 12288, 2048),
                              (1000, 3072, 1024), (1000, 1024, 3072), (125, 3072, 1024), (125, 1024, 3072), (37, 192, 512), (8, 2048, 1024)):
                bm, bn, bk = pick(M, N, K, a16)
                assert bm in (32, 64, 128) and bn in (32, 64, 128) and bk in (128, 256) and N % bn == 0 and K % bk == 0, (M, N, K, a16, bm, bn, bk)
                assert bk == 128 or (bm == 64 and (a16 or bn == 64)) or bm == 32, (M, N, K, a16, bm, bn, bk)
                assert (bm == 32) <= (M <= 128 and bk == 256 and bn <= 64) and (bn == 32) <= (bm == 32), (M, N, K, a16, bm, bn, bk)    # round 6: 32-row tiles, <= 128 rows only
            for (M, N, K) in ((512, 4096, 2048), (512, 2048, 2048), (512, 2048, 6144), (1000, 1024, 3072), (125, 3072, 1024), (125, 1024, 3072)):
                bm, bn, bk = pick(M, N, K, a16)                              # 128 x 128 would hold 128 workgroups or fewer: at least twice as many
                cd = lambda a, b: -(-a // b)
                assert cd(M, bm) * cd(N, bn) >= 2 * cd(M, 128) * cd(N, 128), ((M, N, K), a16, (bm, bn, bk))
            for shape in ((512, 2048, 2048), (512, 2048, 6144)):
                assert pick(*shape, a16)[:2] == (64, 64), (shape, a16, pick(*shape, a16))     # 64 tiles of 128 x 128: four times as many
            # round 6, at most 128 rows: the codec transformer's o / down at 1 x 10 s take 32 x 32 (16 workgroups of 128 x 128 -> 128; measured
            # 10.1 -> 5.8 and 22.3 -> 12.0 us, profiles/r06_gemm_small_tiles_32row.md), q|k|v a 32-row tile (the SwiGLU launch never takes 32 columns: gate / up pairs need two column tiles per wave)
            assert pick(125, 1024, 3072, a16) == (32, 32, 256) and pick(125, 1024, 1024, a16) == (32, 32, 256)
            assert pick(125, 3072, 1024, a16)[0] == 32
            for shape in ((2048, 4096, 2048), (2048, 12288, 2048)):
                assert pick(*shape, a16)[:2] == (128, 128), (shape, a16, pick(*shape, a16))
            assert pick(512, 12288, 2048, a16)[0] == 128                    # row tiles x 50 MB is the bound: 64-row tiles would double it
        assert pick(512, 2048, 6144, 0) == (64, 64, 256)                     # long K: half the exposed round trips
    finally:
        emu.qtts_debug_gemm_wide_tile(-1)


def test_gemm_tap2_tap_reuse_kernel_real_source(emu):
    """gemm_tap2 (round 2, the codec decoder's bf16 GEMM): bf16 input tile staged once per k-slab with its causal halo and reused
    by every tap, sequence-start zeroing applied in the operand registers (tiles that span two sequences included), k-slabs of
    64 and 96, two LDS buffers, fp32 and / or bf16 output with the consumer's SnakeBeta folded in -- against float64 numpy."""
    g = np.random.default_rng(47)
    cases = [  # M, T, N, K, shifts, act, bias, res, out32, out16, snake16
        (300, 100, 96, 96, [-6, -5, -4, -3, -2, -1, 0], ACT_SNAKE, 1, 0, 0, 1, 0),       # conv7 (d = 1) + act2 -> bf16 only
        (200, 50, 64, 192, [-54, -45, -36, -27, -18, -9, 0], ACT_SNAKE, 1, 0, 0, 1, 0),  # dilation 9: halo 54, 4 sequences in 2 tiles
        (260, 130, 128, 64, [0], ACT_NONE, 1, 1, 1, 1, 1),                               # 1x1 + residual -> fp32 and snake'd bf16
        (150, 75, 192, 128, [-1, 0], ACT_NONE, 1, 0, 1, 1, 1),                           # transposed-conv form (2 taps), N = 2 column tiles
        (129, 43, 96, 288, [-18, -15, -12, -9, -6, -3, 0], ACT_NONE, 0, 0, 1, 0, 0),     # K = 3 slabs of 96, dilation 3
        (40, 40, 64, 64, [-2, -1, 0], ACT_GELU, 1, 1, 1, 0, 0),
    ]
    for (M, T, N, K, shift, act, hb, hr, o32, o16, s16) in cases:
        A = (g.standard_normal((M, K + 8)) * 0.5).astype(np.float32)
        Av, Abits = _bf16_round(A)
        W = (g.standard_normal((len(shift), N, K)) / np.sqrt(K * len(shift))).astype(np.float32)
        Wv, Wbits = _bf16_round(W)
        bias = g.standard_normal(N).astype(np.float32) if hb else None
        res = g.standard_normal((M, N)).astype(np.float32) if hr else None
        ea = np.exp(g.standard_normal(N) * 0.3).astype(np.float32)
        ib = (1 / (np.exp(g.standard_normal(N) * 0.3) + 1e-9)).astype(np.float32)
        ea16 = np.exp(g.standard_normal(N) * 0.3).astype(np.float32)
        ib16 = (1 / (np.exp(g.standard_normal(N) * 0.3) + 1e-9)).astype(np.float32)
        want = _gemm_tap_ref(Av[:, :K], T, Wv, shift, bias, None, res, ea, ib, act)
        want16 = want + ib16 * np.sin(want * ea16) ** 2 if s16 else want
        out = np.full((M, N + 8), 7.0, np.float32)
        out16 = np.full((M, N + 8), 0x4242, np.uint16)
        sh = (C.c_int32 * len(shift))(*shift)
        rc = emu.hostemu_gemm_tap16(Abits.ctypes.data_as(C.c_void_p), K + 8, M, T, _ptr(Wbits), N, K, len(shift), sh,
                                    _ptr(bias) if hb else None, _ptr(res) if hr else None, N, _ptr(ea), _ptr(ib), act,
                                    _ptr(out) if o32 else None, N + 8, out16.ctypes.data_as(C.c_void_p) if o16 else None,
                                    _ptr(ea16) if s16 else None, _ptr(ib16) if s16 else None)
        assert rc == 0, ((M, N, K, shift, act), (emu.qtts_last_error() or b"").decode())
        tol = 2e-3 * max(1.0, float(np.abs(want).max()))
        if o32:
            assert np.abs(out[:, :N] - want).max() <= tol, (M, N, K, shift, act, float(np.abs(out[:, :N] - want).max()))
            assert np.all(out[:, N:] == 7.0)
        if o16:
            got = (out16[:, :N].astype(np.uint32) << 16).view(np.float32)
            assert np.abs(got - want16).max() <= 1e-2 * max(1.0, float(np.abs(want16).max())), (M, N, K, shift, float(np.abs(got - want16).max()))
            assert np.all(out16[:, N:] == 0x4242)


def test_gemm_dma_lds_dma_staged_kernel_real_source(emu, qopt):
    """gemm_dma (round 4): gemm_tap2's arithmetic with both operands staged by LDS-DMA into two buffers -- per-lane source addresses
    that produce the padded 144-byte-row LDS image, 64-wide k-slabs, one barrier per step, the A tile re-staged only when the slab
    changes -- against float64 numpy: 7-tap convolutions at dilation 1 and 9 (halo 54, several sequences per tile, ragged last
    tile), the two-tap transposed-conv form, a plain Linear with residual and both outputs.  (The emulator completes a DMA at
    once: what it checks is addressing and bookkeeping; the barrier / DMA ordering is checked on the MI355X.)"""
    qopt(emu, "QTTS_GEMM_DMA", "1")
    qopt(emu, "QTTS_GEMM_RING", "0")
    _run_gemm_dma_cases(emu, {})


def _run_gemm_dma_cases(emu, keep):
    g = np.random.default_rng(48)
    cases = [  # M, T, N, K, shifts, act, bias, res, out32, out16, snake16
        (300, 100, 128, 128, [-6, -5, -4, -3, -2, -1, 0], ACT_SNAKE, 1, 0, 0, 1, 0),
        (200, 50, 128, 192, [-54, -45, -36, -27, -18, -9, 0], ACT_SNAKE, 1, 0, 0, 1, 0),
        (260, 130, 256, 64, [0], ACT_NONE, 1, 1, 1, 1, 1),
        (150, 75, 256, 128, [-1, 0], ACT_NONE, 1, 0, 1, 1, 1),
        (129, 43, 128, 192, [-18, -15, -12, -9, -6, -3, 0], ACT_NONE, 0, 0, 1, 0, 0),
        (140, 70, 128, 640, [-6, -5, -4, -3, -2, -1, 0], ACT_NONE, 1, 0, 1, 1, 0),     # 10 slabs of 64: the ring wraps several times, slabs alternate buffers
        (130, 130, 128, 512, [0], ACT_NONE, 0, 1, 1, 0, 0),                            # plain Linear, 16 steps of 32
        (64, 64, 128, 64, [-1, 0], ACT_NONE, 0, 0, 1, 0, 0),                           # fewer steps (4) than an 8-deep ring
    ]
    for ci, (M, T, N, K, shift, act, hb, hr, o32, o16, s16) in enumerate(cases):
        A = (g.standard_normal((M, K + 8)) * 0.5).astype(np.float32)
        Av, Abits = _bf16_round(A)
        W = (g.standard_normal((len(shift), N, K)) / np.sqrt(K * len(shift))).astype(np.float32)
        Wv, Wbits = _bf16_round(W)
        bias = g.standard_normal(N).astype(np.float32) if hb else None
        res = g.standard_normal((M, N)).astype(np.float32) if hr else None
        ea = np.exp(g.standard_normal(N) * 0.3).astype(np.float32)
        ib = (1 / (np.exp(g.standard_normal(N) * 0.3) + 1e-9)).astype(np.float32)
        ea16 = np.exp(g.standard_normal(N) * 0.3).astype(np.float32)
        ib16 = (1 / (np.exp(g.standard_normal(N) * 0.3) + 1e-9)).astype(np.float32)
        want = _gemm_tap_ref(Av[:, :K], T, Wv, shift, bias, None, res, ea, ib, act)
        want16 = want + ib16 * np.sin(want * ea16) ** 2 if s16 else want
        out = np.full((M, N + 8), 7.0, np.float32)
        out16 = np.full((M, N + 8), 0x4242, np.uint16)
        sh = (C.c_int32 * len(shift))(*shift)
        rc = emu.hostemu_gemm_tap16(Abits.ctypes.data_as(C.c_void_p), K + 8, M, T, _ptr(Wbits), N, K, len(shift), sh,
                                    _ptr(bias) if hb else None, _ptr(res) if hr else None, N, _ptr(ea), _ptr(ib), act,
                                    _ptr(out) if o32 else None, N + 8, out16.ctypes.data_as(C.c_void_p) if o16 else None,
                                    _ptr(ea16) if s16 else None, _ptr(ib16) if s16 else None)
        assert rc == 0, ((M, N, K, shift, act), (emu.qtts_last_error() or b"").decode())
        tol = 2e-3 * max(1.0, float(np.abs(want).max()))
        if o32:
            assert np.abs(out[:, :N] - want).max() <= tol, (M, N, K, shift, act, float(np.abs(out[:, :N] - want).max()))
            assert np.all(out[:, N:] == 7.0)
        if o16:
            got = (out16[:, :N].astype(np.uint32) << 16).view(np.float32)
            assert np.abs(got - want16).max() <= 1e-2 * max(1.0, float(np.abs(want16).max())), (M, N, K, shift, float(np.abs(got - want16).max()))
            assert np.all(out16[:, N:] == 0x4242)
        keep[ci] = (out.copy(), out16.copy())


def test_gemm_ring_kernel_real_source_bit_identical_to_gemm_dma(emu, qopt):
    """gemm_ring (round 6): gemm_dma's arithmetic on a conflict-free band layout of the LDS tiles, a ring of 4 / 6 / 8 weight tiles of 32 k requested
    ahead, operand fragments double-buffered in registers -- the MFMA sequence of every accumulator is gemm_dma's, so every output (fp32 and bf16, every
    epilogue) is BIT-IDENTICAL to gemm_dma_kernel's, at every ring depth: 7-tap convolutions at dilation 1 / 3 / 9 over several slabs, the two-tap
    transposed form (ring capped at the slab's 4 steps), plain Linears (A tile in the ring), ragged last tiles, fewer steps than stages.  Against float64
    numpy too (the shared case runner).  The emulator completes a DMA at once and runs the waves of a workgroup one after the other between barriers:
    it checks addressing, the cursors and that no buffer is re-requested while a later-running wave still reads it; the counted waits are pinned from
    the ISA (tests/test_host_logic.py) and checked on the MI355X (tests/test_gpu_parity.py)."""
    qopt(emu, "QTTS_GEMM_DMA", "1")
    qopt(emu, "QTTS_GEMM_RING", "0")
    qopt(emu, "QTTS_GEMM_RING_KS", "1")                # (split-K changes the summation order: its own test below)
    want = {}
    _run_gemm_dma_cases(emu, want)
    for ring, nst in (("1", "4"), ("1", "6"), ("1", "8"), ("1", "0"), ("2", "4"), ("2", "8")):
        qopt(emu, "QTTS_GEMM_RING", ring)              # "2": the deep-K plain Linear (case 6; gemm_wide's summation order otherwise) takes the ring too
        qopt(emu, "QTTS_GEMM_RING_NST", nst)
        got = {}
        _run_gemm_dma_cases(emu, got)                  # (every case against float64 numpy)
        for ci in want:
            if ring == "2" and ci == 6: continue
            assert np.array_equal(got[ci][0], want[ci][0]) and np.array_equal(got[ci][1], want[ci][1]), (ring, nst, ci)


def test_gemm_ring_split_k_ordered_combine_real_source(emu, qopt):
    """Round 6: the ring kernel with K split over 2 / 3 / 4 workgroups per tile (grids that leave CUs idle): every workgroup stores its partial accumulators,
    draws a ticket, the one with the last ticket sums the partials in split order and runs the epilogue, and re-arms the counter.  Every case of the
    shared runner (7-tap convolutions over 10 slabs, transposed form, plain Linears, every epilogue) against float64 numpy at each split count the shape
    admits (the launcher lowers a count that does not divide the slabs), twice in a row on the same workspace (the counters re-arm), equal both times, and
    within fp32 summation noise of the unsplit result."""
    qopt(emu, "QTTS_GEMM_DMA", "1")
    qopt(emu, "QTTS_GEMM_RING_NST", "4")
    base = {}
    qopt(emu, "QTTS_GEMM_RING", "2"); qopt(emu, "QTTS_GEMM_RING_KS", "1")
    _run_gemm_dma_cases(emu, base)
    for ks in ("2", "3", "4", "0"):
        qopt(emu, "QTTS_GEMM_RING_KS", ks)
        a, b = {}, {}
        _run_gemm_dma_cases(emu, a)
        _run_gemm_dma_cases(emu, b)
        for ci in base:
            assert np.array_equal(a[ci][0], b[ci][0]) and np.array_equal(a[ci][1], b[ci][1]), (ks, ci)
            assert np.allclose(a[ci][0], base[ci][0], rtol=0, atol=2e-5 * max(1.0, float(np.abs(base[ci][0]).max()))), (ks, ci)


@pytest.mark.parametrize("C,M,T,dil", [(96, 600, 300, 1), (96, 300, 100, 9), (96, 77, 77, 3), (192, 300, 150, 9), (192, 130, 65, 1)])
def test_resunit_fused_kernel_real_source(emu, C, M, T, dil):
    """resunit.hip's fused residual unit (round 3: conv7 -> SnakeBeta -> conv1x1 -> + residual in ONE kernel, weights pre-packed into
    MFMA fragments, conv7's output channels permuted so that its accumulators are the next MFMA's operand) from its real source
    against float64 numpy on bf16-rounded operands: several sequences per tile (causal zero rows inside a tile), partial last
    tile, all three dilations, both channel counts (C = 192 runs the two-wave column split with the LDS half swap)."""
    g = np.random.default_rng(C + M + dil)
    x = g.standard_normal((M, C)).astype(np.float32)
    a1e, a1i = np.exp(0.3 * g.standard_normal(C)).astype(np.float32), (1.0 / (np.exp(0.3 * g.standard_normal(C)) + 1e-9)).astype(np.float32)
    a16f, a16 = _bf16_round((x + a1i * np.sin(x * a1e) ** 2).astype(np.float32))
    W1 = (g.standard_normal((7, C, C)) / np.sqrt(7 * C)).astype(np.float32)
    W2 = (g.standard_normal((C, C)) / np.sqrt(C)).astype(np.float32)
    b1, b2 = (0.1 * g.standard_normal(C)).astype(np.float32), (0.1 * g.standard_normal(C)).astype(np.float32)
    ea2, ib2 = np.exp(0.3 * g.standard_normal(C)).astype(np.float32), (1.0 / (np.exp(0.3 * g.standard_normal(C)) + 1e-9)).astype(np.float32)
    ea16, ib16 = np.exp(0.3 * g.standard_normal(C)).astype(np.float32), (1.0 / (np.exp(0.3 * g.standard_normal(C)) + 1e-9)).astype(np.float32)
    W1r, W2r = _bf16_round(W1)[0].astype(np.float64), _bf16_round(W2)[0].astype(np.float64)
    y1 = np.tile(b1.astype(np.float64), (M, 1))
    for tap in range(7):
        sh = -(6 - tap) * dil
        for m in range(M):
            if (m % T) + sh >= 0:
                y1[m] += W1r[tap] @ a16f[m + sh].astype(np.float64)
    act = _bf16_round((y1 + ib2 * np.sin(y1 * ea2) ** 2).astype(np.float32))[0].astype(np.float64)
    y2 = act @ W2r.T + b2 + x
    want16 = y2 + ib16 * np.sin(y2 * ea16) ** 2
    for with16 in (True, False):
        out = np.full((M, C + 4), 7.0, np.float32)
        c16 = np.zeros((M, C + 4), np.uint16)
        rc = emu.hostemu_resunit(_ptr(a16), C, _ptr(x), C, M, T, dil, C, _ptr(W1), _ptr(b1), _ptr(ea2), _ptr(ib2), _ptr(W2), _ptr(b2),
                                 _ptr(out), C + 4, _ptr(c16), _ptr(ea16) if with16 else None, _ptr(ib16) if with16 else None, None, None)
        assert rc == 0, (emu.qtts_last_error() or b"").decode()
        assert np.all(out[:, C:] == 7.0) and np.all(c16[:, C:] == 0)
        err = float(np.abs(out[:, :C] - y2).max())
        assert err <= 4e-3 * max(1.0, float(np.abs(y2).max())), (C, M, T, dil, err)      # (a bf16 flip of one activation moves a sum by ~1e-3)
        assert float(np.sqrt(((out[:, :C] - y2) ** 2).mean())) <= 4e-4 * float(np.sqrt((y2 ** 2).mean()))
        got16 = (c16[:, :C].astype(np.uint32) << 16).view(np.float32)
        ref16 = want16 if with16 else y2
        assert float(np.abs(got16 - ref16).max()) <= 1.2e-2 * max(1.0, float(np.abs(ref16).max()))
    # the bf16 residual stream inside a decoder block: residual in as bf16 (stride C + 4 like the outputs), stream out as bf16
    xr, xr16 = _bf16_round(x)
    x16p = np.zeros((M, C + 4), np.uint16); x16p[:, :C] = xr16
    r16 = np.zeros((M, C + 4), np.uint16)
    c16 = np.zeros((M, C + 4), np.uint16)
    rc = emu.hostemu_resunit(_ptr(a16), C, None, C + 4, M, T, dil, C, _ptr(W1), _ptr(b1), _ptr(ea2), _ptr(ib2), _ptr(W2), _ptr(b2),
                             None, C + 4, _ptr(c16), _ptr(ea16), _ptr(ib16), _ptr(x16p), _ptr(r16))
    assert rc == 0, (emu.qtts_last_error() or b"").decode()
    y2r = act @ W2r.T + b2 + xr
    gotr = (r16[:, :C].astype(np.uint32) << 16).view(np.float32)
    assert np.all(r16[:, C:] == 0) and float(np.abs(gotr - y2r).max()) <= 1.2e-2 * max(1.0, float(np.abs(y2r).max()))
    got16 = (c16[:, :C].astype(np.uint32) << 16).view(np.float32)
    w16 = y2r + ib16 * np.sin(y2r * ea16) ** 2
    assert float(np.abs(got16 - w16).max()) <= 1.2e-2 * max(1.0, float(np.abs(w16).max()))


@pytest.mark.parametrize("bf16", [0, 1])
def test_skinny_kernel_real_source(emu, bf16):
    """skinny.hip itself (packed weight tiles, LDS-staged activations, in-kernel RMSNorm statistics, the two m-tile
    variant, SwiGLU strip pairs, residual) on the emulator against float64 numpy."""
    g = np.random.default_rng(50 + bf16)
    for (M, N, K, norm, act, hb, hr) in [(1, 64, 64, 0, ACT_NONE, 1, 0), (5, 96, 128, 1, ACT_NONE, 0, 1), (16, 64, 256, 1, ACT_SWIGLU, 0, 1),
                                          (23, 128, 128, 1, ACT_NONE, 0, 1), (32, 64, 192, 1, ACT_SWIGLU, 0, 0), (40, 32, 64, 1, ACT_NONE, 1, 0)]:
        x = g.standard_normal((M, K + 4)).astype(np.float32)
        W = (g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        gw = (1 + 0.1 * g.standard_normal(K)).astype(np.float32) if norm else None
        bias = g.standard_normal(N).astype(np.float32) if hb else None
        No = N // 2 if act == ACT_SWIGLU else N
        res = g.standard_normal((M, No)).astype(np.float32) if hr else None
        Wf = W * gw if norm else W
        if bf16:
            Wf = _bf16_round(Wf)[0]
            xv = _bf16_round(x[:, :K])[0]
        else:
            xv = x[:, :K]
        acc = xv.astype(np.float64) @ Wf.astype(np.float64).T
        if norm:
            acc *= 1 / np.sqrt((x[:, :K].astype(np.float64) ** 2).mean(1, keepdims=True) + 1e-6)
        if hb:
            acc += bias
        if act == ACT_SWIGLU:
            a = acc.reshape(M, N // 32, 2, 16)
            acc = ((a[:, :, 0] / (1 + np.exp(-a[:, :, 0]))) * a[:, :, 1]).reshape(M, No)
        if hr:
            acc = acc + res
        out = np.full((M, No + 4), 7.0, np.float32)
        rc = emu.hostemu_skinny(_ptr(x), K + 4, M, _ptr(W), N, K, _ptr(gw) if norm else None, norm, 1e-6, _ptr(bias) if hb else None,
                                _ptr(res) if hr else None, No, act, _ptr(out), No + 4, bf16)
        assert rc == 0, ((M, N, K), (emu.qtts_last_error() or b"").decode())
        tol = (4e-3 if bf16 else 2e-5) * max(1.0, float(np.abs(acc).max()))
        assert np.abs(out[:, :No] - acc).max() <= tol, (M, N, K, norm, act, float(np.abs(out[:, :No] - acc).max()))
        assert np.all(out[:, No:] == 7.0)


def test_skinny_f32_batch8_kernel_real_source(emu, qopt):
    """skinny8_f32_kernel (round 4: the parity mode's frame-step GEMM at batch <= 8 -- tile pairs, whole-line x requests, DPP-rotated
    odd tiles, in-kernel RMSNorm statistics, three-chunk K = 6144) in every instantiation, against float64 numpy; no ss_in is
    handed over for the normalised cases (tests/hostemu/test_entries.cpp)."""
    g = np.random.default_rng(58)
    cases = [(8, 32, 1024, 1, ACT_NONE, 0, 1, 0), (1, 32, 1024, 1, ACT_SWIGLU, 0, 1, 4), (3, 32, 2048, 1, ACT_SWIGLU, 0, 0, 0),
             (7, 16, 2048, 1, ACT_NONE, 1, 0, 8), (8, 16, 3072, 0, ACT_NONE, 1, 1, 0), (2, 32, 3072, 1, ACT_SWIGLU, 0, 0, 0),
             (6, 16, 3072, 1, ACT_NONE, 0, 1, 16), (8, 16, 6144, 1, ACT_NONE, 0, 1, 0),
             (4, 32, 2048, 0, ACT_SWIGLU, 0, 1, 8), (5, 32, 1024, 1, ACT_SWIGLU, 0, 0, 16)]
    for (M, N, K, norm, act, hb, hr, nw) in cases:
        if nw:
            qopt(emu, "QTTS_SKINNY8F_NW", str(nw))
        else:
            qopt(emu, "QTTS_SKINNY8F_NW", None)
        x = g.standard_normal((M, K + 4)).astype(np.float32)
        W = (g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        gw = (1 + 0.1 * g.standard_normal(K)).astype(np.float32) if norm else None
        bias = g.standard_normal(N).astype(np.float32) if hb else None
        No = N // 2 if act == ACT_SWIGLU else N
        res = g.standard_normal((M, No)).astype(np.float32) if hr else None
        acc = x[:, :K].astype(np.float64) @ (W * gw if norm else W).astype(np.float64).T
        if norm:
            acc *= 1 / np.sqrt((x[:, :K].astype(np.float64) ** 2).mean(1, keepdims=True) + 1e-6)
        if hb:
            acc += bias
        if act == ACT_SWIGLU:
            a = acc.reshape(M, N // 32, 2, 16)
            acc = ((a[:, :, 0] / (1 + np.exp(-a[:, :, 0]))) * a[:, :, 1]).reshape(M, No)
        if hr:
            acc = acc + res
        out = np.full((M + 1, No + 4), 7.0, np.float32)
        rc = emu.hostemu_skinny(_ptr(x), K + 4, M, _ptr(W), N, K, _ptr(gw) if norm else None, norm, 1e-6, _ptr(bias) if hb else None,
                                _ptr(res) if hr else None, No, act, _ptr(out), No + 4, 0)
        assert rc == 0, ((M, N, K), (emu.qtts_last_error() or b"").decode())
        err = float(np.abs(out[:M, :No] - acc).max())
        assert err <= 2e-5 * max(1.0, float(np.abs(acc).max())), (M, N, K, norm, act, nw, err)
        assert np.all(out[:M, No:] == 7.0) and np.all(out[M] == 7.0)


def test_skinny_f32_split_k_producer_and_combining_consumer(emu):
    """Round 4: the fp32 batch <= 8 o- / down-projections split K over two workgroups per strip (half 0 carries the residual) and the
    next GEMM of the chain forms its x as half 0 + half 1 on the way in, writes the combined rows out and
    takes the RMSNorm statistics from them -- against float64 numpy, for the producer K of both stacks (2048, 3072, 6144) and both
    consumer shapes (K = 1024: 8 waves, K = 2048: 16 waves; plain and SwiGLU)."""
    g = np.random.default_rng(77)
    i32, vp = C.c_int32, C.c_void_p
    emu.hostemu_skinny_splitk.argtypes = [vp, i32, vp, i32, i32, vp, vp, i32, vp, C.c_float, i32, vp, vp, vp, i32]
    for (M, K1, N1, N2, act) in [(8, 2048, 1024, 32, ACT_SWIGLU), (5, 3072, 1024, 16, ACT_NONE), (3, 6144, 2048, 32, ACT_SWIGLU),
                                 (8, 2048, 2048, 16, ACT_NONE)]:
        x = g.standard_normal((M, K1)).astype(np.float32)
        W1 = (g.standard_normal((N1, K1)) / np.sqrt(K1)).astype(np.float32)
        res = g.standard_normal((M, N1)).astype(np.float32)
        W2 = (g.standard_normal((N2, N1)) / np.sqrt(N1)).astype(np.float32)
        gw = (1 + 0.1 * g.standard_normal(N1)).astype(np.float32)
        No = N2 // 2 if act == ACT_SWIGLU else N2
        parts = np.full((2, 8, N1), 3.0, np.float32)
        x_out = np.full((M + 1, N1), 7.0, np.float32)
        z = np.full((M + 1, No + 4), 7.0, np.float32)
        rc = emu.hostemu_skinny_splitk(_ptr(x), M, _ptr(W1), N1, K1, _ptr(res), _ptr(W2), N2, _ptr(gw), 1e-6, act, _ptr(parts), _ptr(x_out), _ptr(z), No + 4)
        assert rc == 0, ((M, K1, N1), rc, (emu.qtts_last_error() or b"").decode())
        y64 = x.astype(np.float64) @ W1.astype(np.float64).T
        h0 = x[:, :K1 // 2].astype(np.float64) @ W1[:, :K1 // 2].astype(np.float64).T
        assert np.abs(parts[0, :M] - (res + h0)).max() <= 2e-5 and np.abs(parts[0, :M] + parts[1, :M] - (res + y64)).max() <= 4e-5
        assert np.all(parts[:, M:] == 3.0)                                   # rows >= M are nobody's
        h = res.astype(np.float64) + y64
        assert np.abs(x_out[:M] - h).max() <= 4e-5 and np.all(x_out[M] == 7.0)
        # bit-level: the combined row IS (res + half 0) + half 1 in fp32, in that order (half 0 arrives with the residual in it)
        assert np.array_equal(x_out[:M], parts[0, :M] + parts[1, :M])
        acc = (h / np.sqrt((h ** 2).mean(1, keepdims=True) + 1e-6)) @ (W2 * gw).astype(np.float64).T
        if act == ACT_SWIGLU:
            a = acc.reshape(M, N2 // 32, 2, 16)
            acc = ((a[:, :, 0] / (1 + np.exp(-a[:, :, 0]))) * a[:, :, 1]).reshape(M, No)
        err = float(np.abs(z[:M, :No] - acc).max())
        assert err <= 3e-5 * max(1.0, float(np.abs(acc).max())), (M, K1, N1, N2, act, err)
        assert np.all(z[:M, No:] == 7.0) and np.all(z[M] == 7.0)


def test_skinny_bf16_kernel_frame_step_shapes(emu, qopt):
    """skinny2_kernel the way the frame step launches it: x as the producer's bf16 copy, the real models' K (every wave owns the
    same number of k-tiles: the branch-free EXACT instantiations, one / two / three chunks), narrow strips (4 / 8 / 16
    features), M <= 16 / 32 / 64 rows, row variances from the X.X^T MFMA, bf16 shadow output -- against float64 numpy."""
    g = np.random.default_rng(61)
    cases = [(8, 64, 1024, 16, 1, ACT_NONE, 0, 1, 1),     # cp qkv-like: one chunk of 4
             (16, 32, 2048, 4, 0, ACT_NONE, 0, 1, 1),     # cp o-proj-like: 4-feature strips, 2 tokens x 8 rows
             (8, 32, 3072, 4, 0, ACT_NONE, 0, 1, 0),      # cp down-like: three chunks of 4
             (8, 64, 2048, 16, 1, ACT_SWIGLU, 0, 0, 0),   # gate/up strip pairs
             (8, 32, 6144, 8, 0, ACT_NONE, 1, 1, 0),      # talker down: three chunks of 8, 8-feature strips
             (24, 64, 1024, 16, 1, ACT_NONE, 0, 0, 1),    # two m-tiles
             (40, 32, 2048, 8, 1, ACT_NONE, 0, 1, 0),     # four m-tiles
             # round 3: batch 17..32 through the straight-line instantiations (1..3 chunks of 4 k-tiles; strip pairs: 2 / 4 chunks of 2)
             (32, 32, 2048, 8, 0, ACT_NONE, 0, 1, 1),     # cp o-proj at batch 32: two chunks, 8-feature strips, residual + shadow
             (20, 32, 3072, 8, 0, ACT_NONE, 1, 1, 0),     # cp down at batch 20: three chunks, ragged second m-tile
             (32, 64, 1024, 16, 1, ACT_SWIGLU, 0, 0, 0),  # cp gate|up at batch 32: strip pairs, two chunks of 2
             (27, 64, 2048, 16, 1, ACT_SWIGLU, 0, 1, 0),  # talker gate|up: strip pairs, four chunks of 2
             (3, 48, 160, 16, 1, ACT_NONE, 1, 0, 0),      # odd K: generic (guarded) instantiation, 4 waves
             # skinny8_kernel (batch <= 8: tile pairs, whole-line x requests, DPP-rotated odd tiles) beyond the cases above
             (5, 32, 2048, 8, 1, ACT_NONE, 1, 1, 1),      # 5 rows (rows 5..7 re-read row 0), 8-feature strips, norm + bias + res + shadow
             (8, 32, 3072, 8, 0, ACT_NONE, 0, 1, 0),      # six pairs per wave, rotated weights
             (1, 64, 3072, 16, 1, ACT_SWIGLU, 0, 1, 0),   # one row, strip pairs, six pairs per wave
             (8, 48, 2048, 16, 0, ACT_NONE, 1, 0, 1),     # talker o-proj-like, 16-feature strips, no norm
             # round 3: SwiGLU in ONE strip (8 gate + 8 up rows; twice the workgroups of the strip pairs), batch <= 8 only
             (8, 64, 2048, 16, 1, ACT_SWIGLU8, 0, 0, 0),  # talker gate|up-like
             (3, 48, 1024, 16, 1, ACT_SWIGLU8, 0, 1, 0)]  # code-predictor-like, 3 rows, with a residual
    for (M, N, K, fs, norm, act, hb, hr, sh), s8 in [(c, v) for c in cases for v in ("1", "0")]:
        if s8 == "0" and (act == ACT_SWIGLU8 or not (M <= 8 and K % 512 == 0 and fs >= 8)):
            continue                                      # (QTTS_SKINNY8=0: the same shapes through skinny2_kernel)
        qopt(emu, "QTTS_SKINNY8", s8)
        x = (g.standard_normal((M, K + 8)) * 0.7).astype(np.float32)
        W = (g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        gw = (1 + 0.1 * g.standard_normal(K)).astype(np.float32) if norm else None
        bias = g.standard_normal(N).astype(np.float32) if hb else None
        No = N // 2 if act in (ACT_SWIGLU, ACT_SWIGLU8) else N
        res = g.standard_normal((M, No)).astype(np.float32) if hr else None
        Wf = _bf16_round(W * gw if norm else W)[0]
        xv = _bf16_round(x[:, :K])[0]
        acc = xv.astype(np.float64) @ Wf.astype(np.float64).T
        if norm:            # the variance is taken from the bf16 image (what the reference's bf16 path sees)
            acc *= 1 / np.sqrt((xv.astype(np.float64) ** 2).mean(1, keepdims=True) + 1e-6)
        if hb:
            acc += bias
        if act in (ACT_SWIGLU, ACT_SWIGLU8):                  # W rows: blocks of 16 (8) gate rows, then 16 (8) up rows of the same columns
            blk = 16 if act == ACT_SWIGLU else 8
            a = acc.reshape(M, N // (2 * blk), 2, blk)
            acc = ((a[:, :, 0] / (1 + np.exp(-a[:, :, 0]))) * a[:, :, 1]).reshape(M, No)
        if hr:
            acc = acc + res
        out = np.full((M, No + 4), 7.0, np.float32)
        out16 = np.full((M, No + 4), 0x4242, np.uint16)
        rc = emu.hostemu_skinny_bf16x(_ptr(x), K + 8, M, _ptr(W), N, K, _ptr(gw) if norm else None, norm, 1e-6,
                                      _ptr(bias) if hb else None, _ptr(res) if hr else None, No, act, _ptr(out), No + 4, fs,
                                      out16.ctypes.data_as(C.c_void_p) if sh else None)
        assert rc == 0, ((M, N, K, fs), (emu.qtts_last_error() or b"").decode())
        tol = 2e-3 * max(1.0, float(np.abs(acc).max()))
        assert np.abs(out[:, :No] - acc).max() <= tol, (M, N, K, fs, norm, act, float(np.abs(out[:, :No] - acc).max()))
        assert np.all(out[:, No:] == 7.0), "wrote outside its columns"
        if sh:
            got = (out16[:, :No].astype(np.uint32) << 16).view(np.float32)
            assert np.abs(got - out[:, :No]).max() <= 8e-3 * max(1.0, float(np.abs(out).max())), "bf16 shadow differs from the fp32 output"
            assert np.all(out16[:, No:] == 0x4242)


def test_skinny_bf16_split_k_batch_17_to_32(emu, qopt):
    """Round 6: `skinny2_ks_kernel` -- the o- and down-projections at batch 17..32 with K split over the workgroups of a 32-feature strip group
    and the partial sums combined inside the launch (tagged granules, the group's last workgroup adds them in k order) -- in its four
    instantiations against float64 numpy and, bit for bit, against `skinny2_kernel` regrouped: the split only changes WHERE the k-tiles of a
    wave are added, so |split - unsplit| stays within fp32 rounding of the sums.  Two launches on one workspace (the second finds the first's
    granules under another tag), ragged second m-tile, rows >= M and columns >= N untouched, K below QTTS_SKINNY_KS_MINK not split."""
    g = np.random.default_rng(606)
    i32, vp = C.c_int32, C.c_void_p
    emu.hostemu_skinny_ksplit.argtypes = [vp, i32, i32, vp, i32, i32, vp, vp, i32, vp, i32, i32, vp, i32, i32, vp]
    cases = [(32, 2048, 6144, 8, 0, 1, 1),      # talker down 1.7B: 4 strips x 8 features, KS = 4, six k-tiles per wave
             (19, 1024, 3072, 8, 1, 1, 1),      # down of the 1024-wide stacks: KS = 8, four waves x three tiles, ragged second m-tile, bias
             (32, 2048, 2048, 16, 0, 1, 0),     # talker o 1.7B: 2 strips x 16, KS = 4
             (25, 1024, 2048, 8, 0, 0, 1)]      # o of the 1024-wide stacks: KS = 8, one tile per wave, no residual
    qopt(emu, "QTTS_SKINNY_KS_MINK", "1024")
    for (M, N, K, fs, hb, hr, sh) in cases:
        x = (g.standard_normal((M, K + 8)) * 0.7).astype(np.float32)
        W = (g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        bias = g.standard_normal(N).astype(np.float32) if hb else None
        res = g.standard_normal((M, N)).astype(np.float32) if hr else None
        acc = _bf16_round(x[:, :K])[0].astype(np.float64) @ _bf16_round(W)[0].astype(np.float64).T
        if hb:
            acc += bias
        if hr:
            acc = acc + res
        outs = {}
        for ks_on in ("1", "0"):
            qopt(emu, "QTTS_SKINNY_KS", ks_on)
            out = np.full((M + 1, N + 4), 7.0, np.float32)
            out16 = np.full((M + 1, N + 4), 0x4242, np.uint16)
            took = C.c_int32(-1)
            rc = emu.hostemu_skinny_ksplit(_ptr(x), K + 8, M, _ptr(W), N, K, _ptr(bias) if hb else None, _ptr(res) if hr else None, N, _ptr(out), N + 4, fs,
                                           out16.ctypes.data_as(vp) if sh else None, 2, 5, C.byref(took))
            assert rc == 0 and took.value == 1, ((M, N, K, fs), rc, took.value, (emu.qtts_last_error() or b"").decode())
            tol = 2e-3 * max(1.0, float(np.abs(acc).max()))
            assert np.abs(out[:M, :N] - acc).max() <= tol, (M, N, K, fs, ks_on, float(np.abs(out[:M, :N] - acc).max()))
            assert np.all(out[:M, N:] == 7.0) and np.all(out[M] == 7.0), "wrote outside its rows / columns"
            if sh:
                got = (out16[:M, :N].astype(np.uint32) << 16).view(np.float32)
                assert np.abs(got - out[:M, :N]).max() <= 8e-3 * max(1.0, float(np.abs(out).max()))
                assert np.all(out16[:M, N:] == 0x4242) and np.all(out16[M] == 0x4242)
            outs[ks_on] = out[:M, :N].copy()
        d = float(np.abs(outs["1"] - outs["0"]).max())
        assert 0.0 < d <= 2e-5 * max(1.0, float(np.abs(acc).max())), ("split and unsplit sums differ by more than fp32 regrouping (or the split did not run)", d)
    # below the K floor the launcher keeps skinny2_kernel (bit-identical to QTTS_SKINNY_KS=0)
    qopt(emu, "QTTS_SKINNY_KS_MINK", None)       # (default 6144)
    (M, N, K, fs) = (32, 1024, 3072, 8)
    x = (g.standard_normal((M, K + 8)) * 0.7).astype(np.float32)
    W = (g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    res = g.standard_normal((M, N)).astype(np.float32)
    o = {}
    for ks_on in ("1", "0"):
        qopt(emu, "QTTS_SKINNY_KS", ks_on)
        out = np.zeros((M, N), np.float32)
        took = C.c_int32(-1)
        assert emu.hostemu_skinny_ksplit(_ptr(x), K + 8, M, _ptr(W), N, K, None, _ptr(res), N, _ptr(out), N, fs, None, 1, 9, C.byref(took)) == 0
        o[ks_on] = out
    assert np.array_equal(o["1"], o["0"])


def test_prefill_attention_four_rows_per_wave_is_bit_identical_to_one_row_per_wave(emu, qopt):
    """Round 6: `attn_rows4_kernel` (attention.hip) -- the talker prefill's full-causal attention with four consecutive query rows of a
    (sequence, head) per wave, so that a trip's K / V rows serve four rows (a batch-32 prefill is 32 768 one-row waves per layer: 91 us on the
    MI355X).  Without a window every row of a sequence starts at the same key, so per row nothing changes: the output must equal
    `attn_rows_kernel`'s (QTTS_ATTN_ROWS4=0) BIT FOR BIT -- fp32 rows and the bf16 image -- for ragged left-padded batches, T not a multiple
    of 4, head_dim 128 and 64, GQA 2:1, row blocks that start inside the pad; and float64 numpy to fp32 accuracy.  Windowed attention (the
    codec's transformer) keeps the one-row kernel (the switch changes nothing there)."""
    g = np.random.default_rng(414)
    i32, vp = C.c_int32, C.c_void_p
    emu.hostemu_attn_rows.argtypes = [vp, i32, i32, i32, i32, i32, vp, i32, vp, vp]
    for (B, T, nh, nkv, hd, pads, window) in [(3, 21, 4, 2, 128, [0, 5, 18], 0), (2, 9, 4, 2, 64, [2, 0], 0), (1, 37, 2, 2, 128, None, 0), (2, 24, 4, 4, 64, [0, 3], 10)]:
        qkv = g.standard_normal((B, T, (nh + 2 * nkv) * hd)).astype(np.float32)
        npad = np.array(pads, np.int32) if pads is not None else None
        res = {}
        for mode in ("1", "0"):
            qopt(emu, "QTTS_ATTN_ROWS4", mode)
            for with16 in (0, 1):
                out = np.full((B, T, nh * hd), 7.0, np.float32)
                out16 = np.full((B, T, nh * hd), 0x4242, np.uint16)
                rc = emu.hostemu_attn_rows(_ptr(qkv), B, T, nh, nkv, hd, npad.ctypes.data_as(vp) if npad is not None else None, window, _ptr(out),
                                           out16.ctypes.data_as(vp) if with16 else None)
                assert rc == 0, (emu.qtts_last_error() or b"").decode()
                res[(mode, with16)] = (out16 if with16 else out).copy()
        assert np.array_equal(res[("1", 0)], res[("0", 0)]) and np.array_equal(res[("1", 1)], res[("0", 1)]), (B, T, nh, hd, window)
        # float64 reference (rows inside the pad are zeros)
        q = qkv[..., :nh * hd].reshape(B, T, nh, hd).astype(np.float64)
        k = qkv[..., nh * hd:(nh + nkv) * hd].reshape(B, T, nkv, hd).astype(np.float64)
        v = qkv[..., (nh + nkv) * hd:].reshape(B, T, nkv, hd).astype(np.float64)
        ref = np.zeros((B, T, nh, hd))
        for b in range(B):
            p0 = int(npad[b]) if npad is not None else 0
            for t in range(p0, T):
                lo = max(p0, t - window + 1) if window > 0 else p0
                for h in range(nh):
                    kh = h // (nh // nkv)
                    sc = (k[b, lo:t + 1, kh] @ q[b, t, h]) / np.sqrt(hd)
                    w = np.exp(sc - sc.max()); w /= w.sum()
                    ref[b, t, h] = w @ v[b, lo:t + 1, kh]
        assert float(np.abs(res[("1", 0)].reshape(B, T, nh, hd) - ref).max()) <= 2e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("bf16", [0, 1])
def test_attn_decode_kernel_real_source_long_sequences(emu, bf16):
    """attention.hip's decode kernel from its real source against float64 numpy: q/k RMSNorm + rotate-half RoPE at position
    S0 + t - n_pad, K/V append through the cache type, left-pad and causal masks, GQA 2:1, for cache lengths on both sides of
    the register-prefetch window (<= 256 keys bf16 / 128 fp32) -- the multi-round tail loop is what a long utterance runs --
    with one and two new tokens, a permuted page table, and bit-identical results across wave scheduling orders."""
    g = np.random.default_rng(77 + bf16)
    HD, nh, nkv, eps = 128, 4, 2, 1e-6
    inv_freq = (1.0 / (10000.0 ** (np.arange(64) / 64.0))).astype(np.float32)
    qw = (1 + 0.1 * g.standard_normal(HD)).astype(np.float32)
    kw = (1 + 0.1 * g.standard_normal(HD)).astype(np.float32)

    def rnd(a):
        return _bf16_round(a)[0] if bf16 else a

    def normrope(x, w, pos):
        x = x.astype(np.float64)
        x = w * (x / np.sqrt((x ** 2).mean() + eps))
        ang = np.float32(pos) * inv_freq                     # fp32 angle like the kernel, then exact cos / sin
        c, s = np.cos(ang.astype(np.float64)), np.sin(ang.astype(np.float64))
        return np.concatenate([x[:64] * c - x[64:] * s, x[64:] * c + x[:64] * s])

    # npads None + S0 <= 15: the code predictor's call shape (static length, no padding, 32-slot score buffer)
    for (B, n_new, S0, npads, permute) in [(2, 1, 37, [0, 5], False), (3, 2, 130, [0, 17, 64], False), (2, 1, 300, [3, 0], True),
                                           (2, 2, 701, [0, 40], False), (1, 1, 1030, [9], True),
                                           (3, 1, 1, None, False), (2, 1, 7, None, True), (3, 1, 15, None, False), (3, 2, 0, None, False),
                                           (2, 1, 200, [0, 31], True), (3, 1, 252, [5, 0, 100], False), (2, 1, 16, [0, 15], False)]:
        pps = (S0 + n_new + 15) // 16 + 1
        n_pages = B * pps
        table = np.arange(n_pages, dtype=np.int32).reshape(B, pps)
        if permute:
            table = g.permutation(n_pages).astype(np.int32).reshape(B, pps)
        ld = (nh + 2 * nkv) * HD
        qkv = g.standard_normal((n_new * B, ld)).astype(np.float32)
        K = rnd((g.standard_normal((B, nkv, S0, HD)) * 0.7).astype(np.float32))
        V = rnd(g.standard_normal((B, nkv, S0, HD)).astype(np.float32))
        kp = np.full((n_pages, nkv, 16, HD), np.nan, np.float32)      # never-written slots must never be read into the result
        vp_ = np.full((n_pages, nkv, 16, HD), np.nan, np.float32)
        cp_call = npads is None
        npad = np.zeros(B, np.int32) if cp_call else np.asarray(npads, np.int32)
        for b in range(B):
            for s in range(npad[b], S0):                                # left-pad slots stay unwritten (as after a real prefill)
                kp[table[b, s // 16], :, s % 16] = K[b, :, s]
                vp_[table[b, s // 16], :, s % 16] = V[b, :, s]
        if bf16:
            kpool, vpool = _bf16_round(np.nan_to_num(kp, nan=0.0))[1].copy(), _bf16_round(np.nan_to_num(vp_, nan=0.0))[1].copy()
            nanmask = np.isnan(kp)
            kpool[nanmask] = 0x7FC0; vpool[nanmask] = 0x7FC0            # bf16 NaN
        else:
            kpool, vpool = kp.copy(), vp_.copy()
        # ---- float64 reference
        ref = np.zeros((n_new * B, nh * HD))
        newk = np.zeros((B, nkv, n_new, HD)); newv = np.zeros((B, nkv, n_new, HD))
        for b in range(B):
            for h in range(nkv):
                for t in range(n_new):
                    row = qkv[t * B + b]
                    newk[b, h, t] = rnd(normrope(row[(nh + h) * HD:(nh + h + 1) * HD], kw, S0 + t - npad[b]).astype(np.float32))
                    newv[b, h, t] = rnd(row[(nh + nkv + h) * HD:(nh + nkv + h + 1) * HD])
                keys = np.concatenate([K[b, h].astype(np.float64), newk[b, h]], 0)
                vals = np.concatenate([V[b, h].astype(np.float64), newv[b, h]], 0)
                for t in range(n_new):
                    for gq in range(nh // nkv):
                        hq = h * (nh // nkv) + gq
                        q = normrope(qkv[t * B + b][hq * HD:(hq + 1) * HD], qw, S0 + t - npad[b])
                        sc = keys @ q / np.sqrt(HD)
                        sidx = np.arange(S0 + n_new)
                        sc[(sidx < npad[b]) | (sidx > S0 + t)] = -np.inf
                        pr = np.exp(sc - sc.max()); pr /= pr.sum()
                        ref[t * B + b, hq * HD:(hq + 1) * HD] = pr @ np.where(np.isfinite(sc)[:, None], vals, 0.0)
        outs = []
        for order in (0, 1, 2):
            kk, vv = kpool.copy(), vpool.copy()
            out = np.full((n_new * B, nh * HD + 4), 5.0, np.float32)
            emu.hostemu_set_fiber_order(order)
            try:
                rc = emu.hostemu_attn_decode(_ptr(qkv), ld, B, n_new, nh, nkv, _ptr(qw), _ptr(kw), eps, _ptr(inv_freq),
                                             None if cp_call else _ptr(npad), S0, _ptr(kk), _ptr(vv), _ptr(table) if permute else None,
                                             pps, bf16, _ptr(out), nh * HD + 4, 32 if cp_call else S0 + n_new + 3)
            finally:
                emu.hostemu_set_fiber_order(0)
            assert rc == 0, ((B, n_new, S0), (emu.qtts_last_error() or b"").decode())
            outs.append(out)
            err = float(np.abs(out[:, :nh * HD] - ref).max())
            assert err <= 3e-5 * max(1.0, float(np.abs(ref).max())), (B, n_new, S0, bf16, order, err)
            assert np.all(out[:, nh * HD:] == 5.0)
            for b in range(B):                                          # the new K / V rows landed in the cache, rounded once
                for t in range(n_new):
                    s = S0 + t
                    gotk = kk[table[b, s // 16], :, s % 16]
                    gotk = (gotk.astype(np.uint32) << 16).view(np.float32) if bf16 else gotk
                    assert np.abs(gotk - newk[b, :, t]).max() <= (2e-2 if bf16 else 1e-5), (b, t)
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_cp_attn_o_fused_launch_real_source(emu):
    """attention.hip's `cp_attn_o_kernel` (round 4: the code predictor's attention AND o-projection of a single-token pass in one
    launch -- the GEMM split over k by kv head, 64 workgroups of 16 waves publishing 8 x 128 partial sums as tagged 8-byte granules, the
    workgroup of the last kv head adding them in kv-head order + the residual) from its real source at the code predictor's real
    dimensions (16 / 8 heads of 128, hidden 1024), against (a) the two launches it replaces -- `attn_cp_kernel` + the decode GEMM -- on
    the same inputs: the K / V rows appended to the cache are BIT-identical (the attention stage is attn_cp's arithmetic statement for
    statement), the hidden rows agree to fp32 summation order (the same bf16 products, 16 partial sums of 128 instead of the decode
    GEMM's tile order); (b) float64 numpy; for three fiber orders, batch 8 / 5 / 1, cache lengths 1..15, contiguous and permuted page
    tables, with and without the RoPE table; every launch runs TWICE on the same partial-sum buffers (the entry point checks that the
    epoch advanced and that no reducer gave up: a second launch may not take the first one's granules), from different starting
    epochs.  (The emulator runs workgroups one after the other in ascending order, the order the device dispatches them in; a
    reducer that had to WAIT for a producer is what only the hardware runs -- the GPU suite's run-to-run comparison covers it.)"""
    g = np.random.default_rng(404)
    HD, nh, nkv, H, eps = 128, 16, 8, 1024, 1e-6
    qd, ld = nh * HD, (nh + 2 * nkv) * HD
    inv_freq = (1.0 / (10000.0 ** (np.arange(64) / 64.0))).astype(np.float32)
    qw = (1 + 0.1 * g.standard_normal(HD)).astype(np.float32)
    kw = (1 + 0.1 * g.standard_normal(HD)).astype(np.float32)
    Wo = (g.standard_normal((H, qd)) * 0.03).astype(np.float32)
    Wo_r = _bf16_round(Wo)[0].astype(np.float64)
    rope = np.zeros((17, 2, 64), np.float32)                       # launch_rope_table's layout, from the same fp32 angle
    for pos in range(17):
        ang = (np.float32(pos) * inv_freq).astype(np.float32)
        rope[pos, 0], rope[pos, 1] = np.cos(ang), np.sin(ang)

    def normrope(x, w, pos):
        x = x.astype(np.float64)
        x = w * (x / np.sqrt((x ** 2).mean() + eps))
        ang = np.float32(pos) * inv_freq
        c, s = np.cos(ang.astype(np.float64)), np.sin(ang.astype(np.float64))
        return np.concatenate([x[:64] * c - x[64:] * s, x[64:] * c + x[:64] * s])

    for (B, S0, permute, use_tab) in [(8, 7, False, True), (5, 15, True, False), (1, 1, False, True), (8, 2, True, True)]:
        pps = 2
        n_pages = B * pps
        table = (g.permutation(n_pages) if permute else np.arange(n_pages)).astype(np.int32).reshape(B, pps)
        qkv = g.standard_normal((B, ld)).astype(np.float32)
        res = g.standard_normal((B, H)).astype(np.float32)
        K = _bf16_round((g.standard_normal((B, nkv, S0, HD)) * 0.7).astype(np.float32))[0]
        V = _bf16_round(g.standard_normal((B, nkv, S0, HD)).astype(np.float32))[0]
        kpool = np.full((n_pages, nkv, 16, HD), 0x7FC0, np.uint16)  # never-written slots hold bf16 NaN
        vpool = kpool.copy()
        for b in range(B):
            for s in range(S0):
                kpool[table[b, s // 16], :, s % 16] = _bf16_round(K[b, :, s])[1]
                vpool[table[b, s // 16], :, s % 16] = _bf16_round(V[b, :, s])[1]
        # float64 reference: attention (bf16 K / V, new key through bf16), output through bf16, o-projection with bf16 weights
        ref = np.zeros((B, H))
        for b in range(B):
            att = np.zeros(qd)
            for h in range(nkv):
                row = qkv[b]
                nk = _bf16_round(normrope(row[(nh + h) * HD:(nh + h + 1) * HD], kw, S0).astype(np.float32))[0]
                nv = _bf16_round(row[(nh + nkv + h) * HD:(nh + nkv + h + 1) * HD])[0]
                keys = np.concatenate([K[b, h].astype(np.float64), nk[None].astype(np.float64)], 0)
                vals = np.concatenate([V[b, h].astype(np.float64), nv[None].astype(np.float64)], 0)
                for gq in range(2):
                    hq = 2 * h + gq
                    q = normrope(row[hq * HD:(hq + 1) * HD], qw, S0)
                    sc = keys @ q / np.sqrt(HD)
                    pr = np.exp(sc - sc.max()); pr /= pr.sum()
                    att[hq * HD:(hq + 1) * HD] = pr @ vals
            ref[b] = Wo_r @ _bf16_round(att.astype(np.float32))[0].astype(np.float64) + res[b]

        def run(fused, fiber_order=0, epoch0=0):
            kk, vv = kpool.copy(), vpool.copy()
            out = np.full((B, H), np.nan, np.float32)
            out16 = np.full((B, H), 0x4242, np.uint16)
            emu.hostemu_set_fiber_order(fiber_order)
            try:
                rc = emu.hostemu_cp_attn_o(_ptr(qkv), ld, B, _ptr(qw), _ptr(kw), eps, _ptr(inv_freq), S0, _ptr(kk), _ptr(vv),
                                           _ptr(table) if permute else None, pps, _ptr(Wo), H, _ptr(res), _ptr(out), _ptr(out16), fused, 8,
                                           _ptr(rope) if use_tab else None, 17 if use_tab else 0, epoch0)
            finally:
                emu.hostemu_set_fiber_order(0)
            assert rc == 0, ((B, S0, fused), rc, (emu.qtts_last_error() or b"").decode())
            return out, out16, kk, vv

        o0, h0, k0, v0 = run(0)
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(o0 - ref).max()) <= 2e-2 * scale, "the unfused pair is off its own reference"
        first = None
        for (fo, bo) in [(0, 0), (1, 5), (2, 0xFFFFFFF0)]:
            o1, h1, k1, v1 = run(1, fo, bo)
            assert np.array_equal(k1, k0) and np.array_equal(v1, v0), ("K / V append differs from attn_cp", B, S0, bo)
            assert float(np.abs(o1 - o0).max()) <= 2e-5 * scale, (B, S0, bo, float(np.abs(o1 - o0).max()))
            assert float(np.abs(o1 - ref).max()) <= 2e-2 * scale
            assert np.array_equal(h1, _bf16_round(o1)[1]), "bf16 copy of the hidden rows"
            if first is None:
                first = o1
            assert np.array_equal(o1, first), ("result depends on the wave order / the epoch", B, S0, bo, fo)


@pytest.mark.parametrize("H,I", [(256, 1024), (1024, 3072)])
def test_cp_mlp32_one_launch_at_batch_up_to_32_real_source(emu, H, I):
    """`cp_mlp32_kernel` (cp_mlp32.hip, round 6): cp_mlp_kernel's construction for batch 9..32 -- both 16-row tiles of the MFMA's batch columns in
    use, granule buffers of 32 rows per XCD, wave m of a workgroup finishing row tile m, four rows per thread in the reduce.  Per row its
    arithmetic is the batch <= 8 kernel's statement for statement, so at the real dimensions (1024 / 3072) and at 256 / 1024 every block of 8
    rows of a batch of 32 / 19 / 8 must equal, BIT FOR BIT, what `cp_mlp_kernel` gives for those 8 rows alone (fp32 rows and their bf16 copy);
    against float64 numpy and the two decode-GEMM launches to the bf16 bars of the test below; three fiber orders, two launches per call on the
    same granule buffers (0xFF-filled at first) under two serials; rows >= B untouched."""
    g = np.random.default_rng(707 + H)
    eps = 1e-6
    gn = (1 + 0.1 * g.standard_normal(H)).astype(np.float32)
    Wg = (g.standard_normal((I, H)) * 0.05).astype(np.float32)
    Wu = (g.standard_normal((I, H)) * 0.05).astype(np.float32)
    Wd = (g.standard_normal((H, I)) * 0.03).astype(np.float32)
    rw = lambda a: _bf16_round(a)[0]
    Wg_r, Wu_r, Wd_r = rw(Wg * gn[None, :]).astype(np.float64), rw(Wu * gn[None, :]).astype(np.float64), rw(Wd).astype(np.float64)
    vp = C.c_void_p
    for B in (32, 19, 8):
        x = g.standard_normal((B, H)).astype(np.float32)
        res = g.standard_normal((B, H)).astype(np.float32)
        x_r = rw(x).astype(np.float64)
        rs = 1.0 / np.sqrt((x_r ** 2).mean(1, keepdims=True) + eps)
        gg, uu = (x_r @ Wg_r.T) * rs, (x_r @ Wu_r.T) * rs
        ref = rw((gg / (1.0 + np.exp(-gg)) * uu).astype(np.float32)).astype(np.float64) @ Wd_r.T + res
        scale = max(1.0, float(np.abs(ref).max()))

        def run32(fiber_order, epoch0):
            out = np.full((B + 1, H), 7.0, np.float32)
            out16 = np.full((B + 1, H), 0x4242, np.uint16)
            emu.hostemu_set_fiber_order(fiber_order)
            try:
                rc = emu.hostemu_cp_mlp32(_ptr(x), B, _ptr(Wg), _ptr(Wu), _ptr(gn), eps, _ptr(Wd), H, I, _ptr(res), _ptr(out), out16.ctypes.data_as(vp), epoch0)
            finally:
                emu.hostemu_set_fiber_order(0)
            assert rc == 0, ((H, I, B), rc, (emu.qtts_last_error() or b"").decode())
            assert np.all(out[B] == 7.0) and np.all(out16[B] == 0x4242), "wrote a row >= B"
            return out[:B], out16[:B]

        first = None
        for (fo, e0) in [(0, 1), (1, 7), (2, 0xFFFFF0)]:
            o, h = run32(fo, e0)
            assert float(np.abs(o - ref).max()) <= 2e-2 * scale, (H, B, float(np.abs(o - ref).max()))
            assert np.array_equal(h, _bf16_round(o)[1]), "bf16 copy of the hidden rows"
            if first is None:
                first = (o, h)
            assert np.array_equal(o, first[0]), ("result depends on the wave order / the epoch", H, B, fo)
        # every block of <= 8 rows through the batch <= 8 kernel: the same bits
        for r0 in range(0, B, 8):
            nb = min(8, B - r0)
            o8 = np.full((nb, H), np.nan, np.float32)
            h8 = np.full((nb, H), 0x4242, np.uint16)
            xb, rb = np.ascontiguousarray(x[r0:r0 + nb]), np.ascontiguousarray(res[r0:r0 + nb])
            rc = emu.hostemu_cp_mlp(_ptr(xb), nb, _ptr(Wg), _ptr(Wu), _ptr(gn), eps, _ptr(Wd), H, I, _ptr(rb), _ptr(o8), _ptr(h8), 3, 5, 0)
            assert rc == 0, (emu.qtts_last_error() or b"").decode()
            assert np.array_equal(o8, first[0][r0:r0 + nb]) and np.array_equal(h8, first[1][r0:r0 + nb]), ("rows", r0, "differ from cp_mlp_kernel's", H, B)
        # ... and the two decode-GEMM launches it replaces (another fp32 summation order)
        o0 = np.full((B, H), np.nan, np.float32)
        h0 = np.full((B, H), 0x4242, np.uint16)
        rc = emu.hostemu_cp_mlp(_ptr(x), B, _ptr(Wg), _ptr(Wu), _ptr(gn), eps, _ptr(Wd), H, I, _ptr(res), _ptr(o0), _ptr(h0), 0, 0, 0)
        assert rc == 0, (emu.qtts_last_error() or b"").decode()
        assert float(np.sqrt(((first[0] - o0) ** 2).mean())) <= 2e-3 * float(np.sqrt((o0 ** 2).mean())), (H, B)


@pytest.mark.parametrize("H,I,f32", [(256, 1024, 0), (1024, 3072, 0), (256, 1024, 1), (1024, 3072, 1)])
def test_cp_mlp_one_launch_real_source(emu, H, I, f32):
    """`cp_mlp_kernel` (cp_mlp.hip, round 5): the code predictor's MLP of a layer -- RMSNorm, gate|up GEMM, SwiGLU, down GEMM, residual --
    as ONE launch: the intermediate vector sliced by XCD, every workgroup a few features of its XCD's slice (phase A), then its 32 output
    features over that slice (phase B), the eight XCD partials added in XCD order by the workgroup of XCD 7 (phase C); values cross between
    workgroups as tagged granules.  Real source at the code predictor's real dimensions (1024 / 3072: 12 features per workgroup, the padded
    MFMA tile rows) and at 256 / 1024 (16 per workgroup), against the two decode-GEMM launches it replaces (same bf16 products, another
    fp32 summation order, the bf16 rounding of the intermediate vector in between) and against float64 numpy; batch 8 / 3, three fiber
    orders, two launches per call on the same granule buffers under two serials.  The entry point then runs the consuming phases alone:
    under the launch's own (serial, slot) they reproduce the result bit for bit, under another slot or serial every granule is stale and
    the consumers give up, raise the error flag and latch the stop flag.  f32 = 1: the instantiation of the exact parity mode (fp32 operators, rows and
    intermediate vector on v_mfma_f32_16x16x4_f32) against the fp32 engines' two launches and float64 numpy to fp32 precision."""
    g = np.random.default_rng(606 + H)
    eps = 1e-6
    gn = (1 + 0.1 * g.standard_normal(H)).astype(np.float32)
    Wg = (g.standard_normal((I, H)) * 0.05).astype(np.float32)
    Wu = (g.standard_normal((I, H)) * 0.05).astype(np.float32)
    Wd = (g.standard_normal((H, I)) * 0.03).astype(np.float32)
    rw = (lambda a: a) if f32 else (lambda a: _bf16_round(a)[0])
    Wg_r = rw(Wg * gn[None, :]).astype(np.float64)
    Wu_r = rw(Wu * gn[None, :]).astype(np.float64)
    Wd_r = rw(Wd).astype(np.float64)
    for B in (8, 3):
        x = g.standard_normal((B, H)).astype(np.float32)
        res = g.standard_normal((B, H)).astype(np.float32)
        x_r = rw(x).astype(np.float64)
        rs = 1.0 / np.sqrt((x_r ** 2).mean(1, keepdims=True) + eps)
        gg, uu = (x_r @ Wg_r.T) * rs, (x_r @ Wu_r.T) * rs
        act = rw((gg / (1.0 + np.exp(-gg)) * uu).astype(np.float32)).astype(np.float64)
        ref = act @ Wd_r.T + res
        tol, rtol = (2e-5, 2e-6) if f32 else (2e-2, 2e-3)

        def run(mode, fiber_order=0, epoch0=0):
            out = np.full((B, H), np.nan, np.float32)
            out16 = np.full((B, H), 0x4242, np.uint16)
            emu.hostemu_set_fiber_order(fiber_order)
            try:
                rc = emu.hostemu_cp_mlp(_ptr(x), B, _ptr(Wg), _ptr(Wu), _ptr(gn), eps, _ptr(Wd), H, I, _ptr(res), _ptr(out), _ptr(out16), mode, epoch0, f32)
            finally:
                emu.hostemu_set_fiber_order(0)
            assert rc == 0, ((H, I, B, mode), rc, (emu.qtts_last_error() or b"").decode())
            return out, out16

        o0, h0 = run(0)
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(o0 - ref).max()) <= tol * scale, "the two launches are off their own reference"
        first = None
        for (fo, e0) in [(0, 1), (1, 7), (2, 0xFFFFF0)]:
            o3, h3 = run(3, fo, e0)
            assert float(np.abs(o3 - ref).max()) <= tol * scale, (H, B, float(np.abs(o3 - ref).max()))
            assert float(np.sqrt(((o3 - o0) ** 2).mean())) <= rtol * float(np.sqrt((o0 ** 2).mean())), (H, B)
            if not f32:
                assert np.array_equal(h3, _bf16_round(o3)[1]), "bf16 copy of the hidden rows"
            if first is None:
                first = o3
            assert np.array_equal(o3, first), ("result depends on the wave order / the epoch", H, B, fo)


@pytest.mark.parametrize("f32", [0, 1])
def test_cp_layer_front_qkv_attention_o_projection_one_launch_real_source(emu, f32):
    """`cp_attn_o_kernel` with the layer's q|k|v GEMM in front (round 4): 256 workgroups, each first computes a 16-feature strip of
    q|k|v = rsqrt(mean x^2 + eps) * W' x (RMSNorm weight folded into W', row variances from the same bf16 x fragments, four k quarters
    added in wave order), hands it over as tagged granules, then attends with the rows it reads back from six other workgroups' strips,
    and goes on as the attention + o-projection launch.  Real source, real dimensions (hidden 1024, 16 / 8 heads of 128), against the
    three launches it replaces (decode GEMM, attn_cp, decode GEMM: agreement to fp32 summation order of the two GEMMs + the bf16
    rounding of K / V / the attention output that a last-bit difference in q|k|v can flip) and against float64 numpy; batch 8 / 3,
    contiguous and permuted page tables, three fiber orders, two launches per call on the same granule buffers under two serials.  The
    entry point then runs the CONSUMING half alone on those buffers: with the launch's own (serial, slot) it finds every granule and
    reproduces the result bit for bit; with another slot or another serial every granule is stale and the consumers give up and raise
    the error flag instead of taking them (what `generate` turns into QTTS_ERR_STATE).
    f32 = 1 (round 5): the F32 instantiation -- the exact parity mode's operators (fp32, k-tiles of 16 on v_mfma_f32_16x16x4_f32), fp32 rows and
    fp32 cache through the same launch, against the fp32 engines' three launches and float64 numpy to fp32 precision."""
    g = np.random.default_rng(505)
    rw = (lambda a_: np.asarray(a_, np.float32)) if f32 else (lambda a_: _bf16_round(np.asarray(a_, np.float32))[0])
    tol, rtol = (3e-5, 3e-6) if f32 else (3e-2, 2e-3)
    HD, nh, nkv, H, eps, eps_in = 128, 16, 8, 1024, 1e-6, 1e-6
    qd, ld = nh * HD, (nh + 2 * nkv) * HD
    inv_freq = (1.0 / (10000.0 ** (np.arange(64) / 64.0))).astype(np.float32)
    qw = (1 + 0.1 * g.standard_normal(HD)).astype(np.float32)
    kw = (1 + 0.1 * g.standard_normal(HD)).astype(np.float32)
    gn = (1 + 0.1 * g.standard_normal(H)).astype(np.float32)
    Wqkv = (g.standard_normal((ld, H)) * 0.04).astype(np.float32)
    Wo = (g.standard_normal((H, qd)) * 0.03).astype(np.float32)
    Wq_r = rw(Wqkv * gn[None, :]).astype(np.float64)
    Wo_r = rw(Wo).astype(np.float64)

    def normrope(x, w, pos):
        x = x.astype(np.float64)
        x = w * (x / np.sqrt((x ** 2).mean() + eps))
        ang = np.float32(pos) * inv_freq
        c, s = np.cos(ang.astype(np.float64)), np.sin(ang.astype(np.float64))
        return np.concatenate([x[:64] * c - x[64:] * s, x[64:] * c + x[:64] * s])

    for (B, S0, permute) in [(8, 6, False), (3, 14, True)]:
        pps = 2
        n_pages = B * pps
        table = (g.permutation(n_pages) if permute else np.arange(n_pages)).astype(np.int32).reshape(B, pps)
        x = g.standard_normal((B, H)).astype(np.float32)
        res = g.standard_normal((B, H)).astype(np.float32)
        K = _bf16_round((g.standard_normal((B, nkv, S0, HD)) * 0.7).astype(np.float32))[0]
        V = _bf16_round(g.standard_normal((B, nkv, S0, HD)).astype(np.float32))[0]
        kpool = np.full((n_pages, nkv, 16, HD), np.nan, np.float32) if f32 else np.full((n_pages, nkv, 16, HD), 0x7FC0, np.uint16)
        vpool = kpool.copy()
        for b in range(B):
            for s in range(S0):
                kpool[table[b, s // 16], :, s % 16] = K[b, :, s] if f32 else _bf16_round(K[b, :, s])[1]
                vpool[table[b, s // 16], :, s % 16] = V[b, :, s] if f32 else _bf16_round(V[b, :, s])[1]
        x_r = rw(x).astype(np.float64)
        ref = np.zeros((B, H))
        for b in range(B):
            row = (Wq_r @ x_r[b]) / np.sqrt((x_r[b] ** 2).mean() + eps_in)
            att = np.zeros(qd)
            for h in range(nkv):
                nk = rw(normrope(row[(nh + h) * HD:(nh + h + 1) * HD], kw, S0))
                nv = rw(row[(nh + nkv + h) * HD:(nh + nkv + h + 1) * HD])
                keys = np.concatenate([K[b, h].astype(np.float64), nk[None].astype(np.float64)], 0)
                vals = np.concatenate([V[b, h].astype(np.float64), nv[None].astype(np.float64)], 0)
                for gq in range(2):
                    hq = 2 * h + gq
                    q = normrope(row[hq * HD:(hq + 1) * HD], qw, S0)
                    sc = keys @ q / np.sqrt(HD)
                    pr = np.exp(sc - sc.max()); pr /= pr.sum()
                    att[hq * HD:(hq + 1) * HD] = pr @ vals
            ref[b] = Wo_r @ rw(att).astype(np.float64) + res[b]

        def run(mode, fiber_order=0, epoch0=0):
            kk, vv = kpool.copy(), vpool.copy()
            out = np.full((B, H), np.nan, np.float32)
            out16 = np.full((B, H), 0x4242, np.uint16)
            emu.hostemu_set_fiber_order(fiber_order)
            try:
                rc = emu.hostemu_cp_layer_front(_ptr(x), B, _ptr(Wqkv), _ptr(gn), eps_in, _ptr(qw), _ptr(kw), eps, _ptr(inv_freq), S0, _ptr(kk),
                                                _ptr(vv), _ptr(table) if permute else None, pps, _ptr(Wo), H, _ptr(res), _ptr(out), _ptr(out16),
                                                mode, None, 0, epoch0, f32)
            finally:
                emu.hostemu_set_fiber_order(0)
            assert rc == 0, ((B, S0, mode), rc, (emu.qtts_last_error() or b"").decode())
            return out, out16, kk, vv

        o0, h0, k0, v0 = run(0)
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(o0 - ref).max()) <= tol * scale, "the three launches are off their own reference"
        first = None
        for (fo, e0) in [(0, 0), (1, 7), (2, 0xFFFFFFF0)]:
            o2, h2, k2, v2 = run(2, fo, e0)
            assert float(np.abs(o2 - ref).max()) <= tol * scale, (B, S0, float(np.abs(o2 - ref).max()))
            assert float(np.sqrt(((o2 - o0) ** 2).mean())) <= rtol * float(np.sqrt((o0 ** 2).mean())), (B, S0)
            if f32:     # the appended K / V rows: the two q|k|v GEMMs' fp32 sums differ in order only
                live = ~np.isnan(k0)
                assert np.array_equal(live, ~np.isnan(k2)) and np.array_equal(~np.isnan(v0), ~np.isnan(v2))
                assert np.abs(k2[live] - k0[live]).max() <= 2e-5 and np.abs(v2[live] - v0[live]).max() <= 2e-5
            else:
                # the appended K / V rows: equal up to a bf16 last-bit flip where the two q|k|v GEMMs' fp32 sums straddle a rounding boundary
                kd = (k2.astype(np.int32) - k0.astype(np.int32)); vd = (v2.astype(np.int32) - v0.astype(np.int32))
                assert np.abs(kd).max() <= 1 and np.abs(vd).max() <= 1 and (kd != 0).mean() < 0.02 and (vd != 0).mean() < 0.02
                assert np.array_equal(h2, _bf16_round(o2)[1]), "bf16 copy of the hidden rows"
            if first is None:
                first = o2
            assert np.array_equal(o2, first), ("result depends on the wave order / the epoch", B, S0, fo)


@pytest.mark.parametrize("nsplit", [1, 3])
def test_attn_tk16_matrix_pipe_kernel_real_source(emu, nsplit, qopt):
    """attention.hip's `attn_tk16_kernel` (round 3: the talker's single-token decode attention with BOTH products on the matrix
    pipe; bf16 cache, V pages stored dim-major [128][16 keys]) from its real source against float64 numpy: q / k RMSNorm + RoPE,
    K row-major / V transposed append, left-pad mask, never-written slots holding NaN, GQA 2:1 and 1:1, cache lengths from 1 key
    to past the 256-key register window, contiguous and permuted page tables, alone and as split-KV partials + merge.  q, K, P, V
    enter the MFMA as bf16 (the precision of the reference's own bf16 attention), so the bar is bf16-sized: 2 % of the largest
    output and 0.4 % RMS; bit-identical across wave scheduling orders."""
    qopt(emu, "QTTS_DEBUG_ATTN_VT", "1")
    if nsplit > 1:
        qopt(emu, "QTTS_DEBUG_ATTN_NSPLIT", str(nsplit))
    g = np.random.default_rng(91 + nsplit)
    HD, eps = 128, 1e-6
    inv_freq = (1.0 / (10000.0 ** (np.arange(64) / 64.0))).astype(np.float32)
    qw = (1 + 0.1 * g.standard_normal(HD)).astype(np.float32)
    kw = (1 + 0.1 * g.standard_normal(HD)).astype(np.float32)
    rnd = lambda a: _bf16_round(a)[0]

    def normrope(x, w, pos):
        x = x.astype(np.float64)
        x = w * (x / np.sqrt((x ** 2).mean() + eps))
        ang = np.float32(pos) * inv_freq
        c, s_ = np.cos(ang.astype(np.float64)), np.sin(ang.astype(np.float64))
        return np.concatenate([x[:64] * c - x[64:] * s_, x[64:] * c + x[:64] * s_])

    for (B, nh, nkv, S0, npads, permute) in [(2, 4, 2, 37, [0, 5], False), (2, 4, 2, 1, [0, 0], False), (3, 4, 2, 130, [0, 17, 64], True),
                                             (2, 2, 2, 200, [3, 0], False), (2, 4, 2, 300, [0, 40], True), (1, 4, 2, 701, [9], False),
                                             (2, 4, 2, 32, [0, 31], False), (2, 4, 2, 255, [0, 100], False)]:
        GQ = nh // nkv
        pps = (S0 + 1 + 15) // 16 + 1
        n_pages = B * pps
        table = g.permutation(n_pages).astype(np.int32).reshape(B, pps) if permute else np.arange(n_pages, dtype=np.int32).reshape(B, pps)
        ld = (nh + 2 * nkv) * HD
        qkv = g.standard_normal((B, ld)).astype(np.float32)
        K = rnd((g.standard_normal((B, nkv, S0, HD)) * 0.7).astype(np.float32))
        V = rnd(g.standard_normal((B, nkv, S0, HD)).astype(np.float32))
        kp = np.full((n_pages, nkv, 16, HD), np.nan, np.float32)      # never-written slots must never reach the result
        vp_ = np.full((n_pages, nkv, HD, 16), np.nan, np.float32)      # V pages: [dim][key]
        npad = np.asarray(npads, np.int32)
        for b in range(B):
            for s_ in range(npad[b], S0):
                kp[table[b, s_ // 16], :, s_ % 16] = K[b, :, s_]
                vp_[table[b, s_ // 16], :, :, s_ % 16] = V[b, :, s_]
        kpool, vpool = _bf16_round(np.nan_to_num(kp, nan=0.0))[1].copy(), _bf16_round(np.nan_to_num(vp_, nan=0.0))[1].copy()
        kpool[np.isnan(kp)] = 0x7FC0; vpool[np.isnan(vp_)] = 0x7FC0
        ref = np.zeros((B, nh * HD))
        newk = np.zeros((B, nkv, HD)); newv = np.zeros((B, nkv, HD))
        for b in range(B):
            for h in range(nkv):
                row = qkv[b]
                newk[b, h] = rnd(normrope(row[(nh + h) * HD:(nh + h + 1) * HD], kw, S0 - npad[b]).astype(np.float32))
                newv[b, h] = rnd(row[(nh + nkv + h) * HD:(nh + nkv + h + 1) * HD])
                keys = np.concatenate([K[b, h].astype(np.float64), newk[b, h][None]], 0)
                vals = np.concatenate([V[b, h].astype(np.float64), newv[b, h][None]], 0)
                for gq in range(GQ):
                    hq = h * GQ + gq
                    q = normrope(row[hq * HD:(hq + 1) * HD], qw, S0 - npad[b])
                    sc = keys @ q / np.sqrt(HD)
                    sc[np.arange(S0 + 1) < npad[b]] = -np.inf
                    pr = np.exp(sc - sc.max()); pr /= pr.sum()
                    ref[b, hq * HD:(hq + 1) * HD] = pr @ np.where(np.isfinite(sc)[:, None], vals, 0.0)
        outs = []
        for order in (0, 1, 2):
            kk, vv = kpool.copy(), vpool.copy()
            out = np.full((B, nh * HD + 4), 5.0, np.float32)
            emu.hostemu_set_fiber_order(order)
            try:
                rc = emu.hostemu_attn_decode(_ptr(qkv), ld, B, 1, nh, nkv, _ptr(qw), _ptr(kw), eps, _ptr(inv_freq), _ptr(npad), S0,
                                             _ptr(kk), _ptr(vv), _ptr(table) if permute else None, pps, 1, _ptr(out), nh * HD + 4, S0 + 4)
            finally:
                emu.hostemu_set_fiber_order(0)
            assert rc == 0, ((B, S0), (emu.qtts_last_error() or b"").decode())
            outs.append(out)
            d = out[:, :nh * HD] - ref
            assert np.isfinite(out).all(), (B, S0, nsplit)
            assert float(np.abs(d).max()) <= 2e-2 * max(1.0, float(np.abs(ref).max())), (B, nh, S0, nsplit, order, float(np.abs(d).max()))
            assert float(np.sqrt((d ** 2).mean())) <= 4e-3 * float(np.sqrt((ref ** 2).mean())) + 1e-4, (B, S0, nsplit)
            assert np.all(out[:, nh * HD:] == 5.0)
            for b in range(B):                                          # the new K row (row-major) and V row (dim-major) landed in the cache
                gotk = (kk[table[b, S0 // 16], :, S0 % 16].astype(np.uint32) << 16).view(np.float32)
                gotv = (vv[table[b, S0 // 16], :, :, S0 % 16].astype(np.uint32) << 16).view(np.float32)
                assert np.abs(gotk - newk[b]).max() <= 2e-2 and np.abs(gotv - newv[b]).max() <= 1e-6, b
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def _ok(lib, rc):
    assert rc == 0, (rc, (lib.qtts_last_error() or b"").decode())


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _codec_emu(emu, dtype):
    c = synth.codec_tiny()
    w = {k: torch.from_numpy(v) for k, v in synth.codec_weights(c).items()}
    cfg = CodecDecoderConfig.from_any(synth.cfg_dict(c))
    cc = _lib.CodecConfigC()
    for f in ("codebook_size", "codebook_dim", "hidden_size", "latent_dim", "num_attention_heads", "num_key_value_heads",
              "head_dim", "sliding_window", "intermediate_size", "num_hidden_layers", "num_quantizers", "decoder_dim"):
        setattr(cc, f, int(getattr(cfg, f)))
    cc.n_upsample_rates, cc.n_upsampling_ratios = len(cfg.upsample_rates), len(cfg.upsampling_ratios)
    for i, r in enumerate(cfg.upsample_rates):
        cc.upsample_rates[i] = int(r)
    for i, r in enumerate(cfg.upsampling_ratios):
        cc.upsampling_ratios[i] = int(r)
    cc.rms_norm_eps, cc.rope_theta = float(cfg.rms_norm_eps), float(cfg.rope_theta)
    cc.compute_dtype, cc.max_batch, cc.max_frames = dtype, 2, 64
    h = C.c_void_p()
    _ok(emu, emu.qtts_codec_create(C.byref(cc), C.byref(h)))
    for name, t in w.items():
        if ".input_proj." in name and name.startswith("quantizer."):
            continue
        _lib.bind_tensor(emu.qtts_codec_bind, h, name, t)
    _ok(emu, emu.qtts_codec_finalize(h))
    return c, w, h


@pytest.fixture(scope="module")
def codec(emu):
    c, w, h = _codec_emu(emu, _lib.QTTS_F32)
    yield c, w, h
    emu.qtts_codec_destroy(h)


def test_decoder_bf16_forward_and_stream(emu):
    """The codec decoder in its serving precision on the emulator (bf16 MFMA GEMMs; round 2: inside the decoder blocks the
    GEMM-only activations travel as bf16 through the tap-reuse kernel, the residual stream stays fp32): within 8 % relative RMS
    of the fp32 oracle on these random weights (measured 4.9 %, the same as with fp32 activations), and the state-carrying
    stream decode -- which keeps fp32 activations between its layers -- stays within 3 % (measured 1.3 %) of the whole-sequence forward."""
    c, w, h = _codec_emu(emu, _lib.QTTS_BF16)
    try:
        T = 8
        codes = np.random.default_rng(3).integers(0, c.codebook_size, (1, c.num_quantizers, T))
        with torch.no_grad():
            ref = codec_ref.decoder_forward(w, c, torch.from_numpy(codes))[:, 0].numpy()
        with real_gemm(emu):
            wav = np.zeros((1, T * c.total_upsample), np.float32)
            _ok(emu, emu.qtts_codec_forward(h, _ptr(codes), 1, T, _ptr(wav), None, None))
            assert np.sqrt(((wav - ref) ** 2).mean()) <= 0.08 * np.sqrt((ref ** 2).mean())
            _ok(emu, emu.qtts_codec_stream_begin(h, 1))
            outs = []
            for a, b in ((0, 1), (1, 4), (4, 5), (5, 8)):
                o = np.zeros((1, (b - a) * c.total_upsample), np.float32)
                _ok(emu, emu.qtts_codec_stream_push(h, _ptr(np.ascontiguousarray(codes[..., a:b])), b - a, _ptr(o), None))
                outs.append(o)
            st_wav = np.concatenate(outs, axis=1)
            assert np.sqrt(((st_wav - wav) ** 2).mean()) <= 0.03 * np.sqrt((wav ** 2).mean())     # measured 1.3 %
    finally:
        emu.qtts_codec_destroy(h)


def test_decoder_orchestration_forward_and_chunked(emu, codec):
    """The validated decoder path, now also under a CPU regression test: finalize() repacking (tap layouts, polyphase
    transposed convs, fused RVQ projection, gate/up interleave) + forward() + chunked decode with -1 padding."""
    c, w, h = codec
    rng = np.random.default_rng(3)
    codes = rng.integers(0, c.codebook_size, (2, c.num_quantizers, 11))
    wav = np.zeros((2, 11 * c.total_upsample), np.float32)
    _ok(emu, emu.qtts_codec_forward(h, _ptr(codes), 2, 11, _ptr(wav), None, None))
    with torch.no_grad():
        ref = codec_ref.decoder_forward(w, c, torch.from_numpy(codes))[:, 0].numpy()
    assert np.sqrt(((wav - ref) ** 2).mean()) <= 1e-5
    with real_gemm(emu):                                                     # and with gemm_tap.hip itself, on a short clip
        w3 = np.zeros((1, 3 * c.total_upsample), np.float32)
        _ok(emu, emu.qtts_codec_forward(h, _ptr(np.ascontiguousarray(codes[:1, :, :3])), 1, 3, _ptr(w3), None, None))
        with torch.no_grad():
            ref3 = codec_ref.decoder_forward(w, c, torch.from_numpy(codes[:1, :, :3]))[:, 0].numpy()
        assert np.sqrt(((w3 - ref3) ** 2).mean()) <= 1e-5
    padded = np.ascontiguousarray(codes.transpose(0, 2, 1)).copy()          # (B, T, Q), row 1 ends after 7 frames
    padded[1, 7:] = -1
    out = np.zeros((2, 11 * c.total_upsample), np.float32)
    lens = (C.c_int64 * 2)()
    _ok(emu, emu.qtts_codec_decode(h, _ptr(padded), 2, 11, 4, 3, _ptr(out), lens, None))
    with torch.no_grad():
        rows = codec_ref.model_decode(w, c, torch.from_numpy(padded))        # default chunking: one chunk here
        chunked = codec_ref.chunked_decode(w, c, torch.clamp(torch.from_numpy(padded), min=0).transpose(1, 2), 4, 3)[:, 0].numpy()
    assert [int(x) for x in lens] == [11 * c.total_upsample, 7 * c.total_upsample] == [r.shape[0] for r in rows]
    assert np.sqrt(((out - chunked) ** 2).mean()) <= 1e-5


def test_decode_call_graph_replay_equals_eager(emu, codec):
    """Round 4: `qtts_codec_forward` / `qtts_codec_decode` replay a captured graph from the second call with the same SHAPE (B, T,
    chunking) on: the engine copies the codes into its own staging buffer, replays, and copies the waveform out, so the caller's
    addresses do not matter (new tensors per call still replay) and new contents are seen; the cache is an LRU of 8 shapes."""
    c, w, h = codec

    class St(C.Structure):
        _fields_ = [("cap", C.c_int32), ("rep", C.c_int32), ("cached", C.c_int32), ("nodes", C.c_int32)]
    emu.qtts_codec_get_stats.argtypes = [C.c_void_p, C.POINTER(St)]

    def stats():
        st = St()
        _ok(emu, emu.qtts_codec_get_stats(h, C.byref(st)))
        return st.cap, st.rep, st.cached, st.nodes
    c0, r0, _, _ = stats()
    rng = np.random.default_rng(31)
    T = 9
    padded = np.ascontiguousarray(rng.integers(0, c.codebook_size, (2, T, c.num_quantizers)))
    out = np.zeros((2, T * c.total_upsample), np.float32)
    lens = (C.c_int64 * 2)()
    refs = []
    for it in range(4):                      # call 0: eager; call 1: capture + replay; calls 2, 3: replay -- each on fresh codes
        padded[...] = rng.integers(0, c.codebook_size, padded.shape)
        if it == 3:
            padded[1, 5:] = -1               # padding rides through the graph like any other code value
        _ok(emu, emu.qtts_codec_decode(h, _ptr(padded), 2, T, 4, 3, _ptr(out), lens, None))
        with torch.no_grad():
            chunked = codec_ref.chunked_decode(w, c, torch.clamp(torch.from_numpy(padded), min=0).transpose(1, 2), 4, 3)[:, 0].numpy()
        assert np.sqrt(((out - chunked) ** 2).mean()) <= 1e-5, it
    cap, rep, cached, nodes = stats()
    assert (cap - c0, rep - r0) == (1, 3) and cached >= 1 and nodes > 20
    assert [int(x) for x in lens] == [T * c.total_upsample, 5 * c.total_upsample]
    # an out-of-range code fails the replayed call too (device flag), and the next call is clean again
    padded[0, 0, 0] = c.codebook_size
    assert emu.qtts_codec_decode(h, _ptr(padded), 2, T, 4, 3, _ptr(out), None, None) != 0
    padded[0, 0, 0] = 0
    _ok(emu, emu.qtts_codec_decode(h, _ptr(padded), 2, T, 4, 3, _ptr(out), None, None))
    # forward(): its own key space (with / without the pre-clamp output); OTHER buffers of the same shape replay the same graph
    codes = np.ascontiguousarray(rng.integers(0, c.codebook_size, (1, c.num_quantizers, 5)))
    w1, w2, w3, pre = (np.zeros((1, 5 * c.total_upsample), np.float32) for _ in range(4))
    _ok(emu, emu.qtts_codec_forward(h, _ptr(codes), 1, 5, _ptr(w1), None, None))
    cap0, rep0 = stats()[:2]
    _ok(emu, emu.qtts_codec_forward(h, _ptr(codes), 1, 5, _ptr(w1), None, None))        # captured here
    codes2 = codes.copy()
    _ok(emu, emu.qtts_codec_forward(h, _ptr(codes2), 1, 5, _ptr(w2), None, None))       # other addresses, same shape: replay
    _ok(emu, emu.qtts_codec_forward(h, _ptr(codes), 1, 5, _ptr(w3), _ptr(pre), None))   # with the pre-clamp output: another key, eager
    cap1, rep1 = stats()[:2]
    assert (cap1 - cap0, rep1 - rep0) == (1, 2)
    assert np.array_equal(w1, w2) and np.array_equal(w1, w3) and np.abs(pre).max() > 0
    # LRU: ten more shapes, each seen twice -> at most 8 graphs stay cached, results still right; growing the staging buffers
    # (a larger shape) drops the graphs that baked the old addresses, and the small shape captures again
    for T2 in range(6, 16):
        cd = np.ascontiguousarray(rng.integers(0, c.codebook_size, (1, c.num_quantizers, T2)))
        a, b = (np.zeros((1, T2 * c.total_upsample), np.float32) for _ in range(2))
        _ok(emu, emu.qtts_codec_forward(h, _ptr(cd), 1, T2, _ptr(a), None, None))
        _ok(emu, emu.qtts_codec_forward(h, _ptr(cd), 1, T2, _ptr(b), None, None))
        assert np.array_equal(a, b)
    assert stats()[2] <= 8
    _ok(emu, emu.qtts_codec_forward(h, _ptr(codes), 1, 5, _ptr(w2), None, None))
    assert np.array_equal(w1, w2)


def test_stream_push_equals_whole_sequence_forward(emu, codec):
    """qtts_codec_stream_begin / _push (state-carrying streaming decode): the product's C++ orchestration,
    run here on the emulated kernels, reproduces the whole-sequence forward for ragged packets, single frames and streams several
    attention windows long."""
    c, w, h = codec
    T = 45
    codes = np.random.default_rng(12).integers(0, c.codebook_size, (2, c.num_quantizers, T))
    with torch.no_grad():
        ref = codec_ref.decoder_forward(w, c, torch.from_numpy(codes))[:, 0].numpy()
    for cuts in ([0, 1, 2, 3, 10, 11, 30, 45], list(range(0, 46, 5)), list(range(20)) + [45]) + (([0, 45], list(range(46))) if FULL else ()):
        _ok(emu, emu.qtts_codec_stream_begin(h, 2))
        outs = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            pk = np.ascontiguousarray(codes[..., a:b])
            o = np.zeros((2, (b - a) * c.total_upsample), np.float32)
            _ok(emu, emu.qtts_codec_stream_push(h, _ptr(pk), b - a, _ptr(o), None))
            outs.append(o)
        got = np.concatenate(outs, axis=1)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 5e-5, cuts[:4]
    with real_gemm(emu):                                                     # gemm_tap.hip itself on the staged rows
        _ok(emu, emu.qtts_codec_stream_begin(h, 1))
        outs = []
        for a, b in ((0, 1), (1, 3), (3, 4)):
            o = np.zeros((1, (b - a) * c.total_upsample), np.float32)
            _ok(emu, emu.qtts_codec_stream_push(h, _ptr(np.ascontiguousarray(codes[:1, :, a:b])), b - a, _ptr(o), None))
            outs.append(o)
        assert np.abs(np.concatenate(outs, axis=1) - ref[:1, :4 * c.total_upsample]).max() <= 5e-5
    assert emu.qtts_codec_stream_begin(h, 3) != 0                            # more sequences than max_batch


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_encoder_orchestration_codes_vs_reference_golden(emu, golden_dir, dtype):
    """qtts_encoder_* (HIP codec encoder): finalize() repacking (stride-1 taps, super-row strided taps,
    normalised codebooks) + encode() on the emulated kernels against the codes of the reference's own encoder class: bit-exact in fp32; in bf16 (the bf16 MFMA kernels, no oracle
    of their own) most codes must still agree."""
    g = np.load(os.path.join(golden_dir, "codec_enc_small.npz"))
    c = synth.mimi_enc_small()
    w = {k: torch.from_numpy(v) for k, v in synth.mimi_enc_weights(c).items()}
    cfg = CodecEncoderConfig.from_any(synth.cfg_dict(c))
    ec = _lib.EncoderConfigC()
    for f in ("hidden_size", "num_filters", "num_residual_layers", "kernel_size", "last_kernel_size", "residual_kernel_size",
              "dilation_growth_rate", "compress", "codebook_size", "codebook_dim", "num_quantizers", "num_semantic_quantizers",
              "num_hidden_layers", "intermediate_size", "num_attention_heads", "num_key_value_heads", "head_dim", "sliding_window"):
        setattr(ec, f, int(getattr(cfg, f)))
    ec.n_ratios = len(cfg.upsampling_ratios)
    for i, r in enumerate(cfg.upsampling_ratios):
        ec.ratios[i] = int(r)
    ec.valid_num_quantizers = cfg.encoder_valid_num_quantizers
    ec.rope_theta, ec.norm_eps = float(cfg.rope_theta), float(cfg.norm_eps)
    ec.compute_dtype, ec.max_batch, ec.max_samples = (_lib.QTTS_F32 if dtype == "f32" else _lib.QTTS_BF16), 2, 512
    h = C.c_void_p()
    _ok(emu, emu.qtts_encoder_create(C.byref(ec), C.byref(h)))
    try:
        for name, t in w.items():
            if not name.endswith("codebook.initialized"):
                _lib.bind_tensor(emu.qtts_encoder_bind, h, name, t)
        _ok(emu, emu.qtts_encoder_finalize(h))
        emu.hostemu_set_real_gemm(1)                                         # gemm_tap.hip itself (restored in `finally`)
        for n in (16, 203, 331):
            x = np.ascontiguousarray(g[f"wav{n}"][:, 0])
            fr = C.c_int64()
            _ok(emu, emu.qtts_encoder_frames(h, n, C.byref(fr)))
            want = g[f"codes{n}"][:, :c.encoder_valid_num_quantizers]
            assert fr.value == want.shape[-1]
            codes = np.zeros((2, c.encoder_valid_num_quantizers, fr.value), np.int64)
            _ok(emu, emu.qtts_encoder_encode(h, _ptr(x), 2, n, _ptr(codes), None))
            if dtype == "f32":
                assert np.array_equal(codes, want), (n, float((codes != want).mean()))
            else:
                assert (codes == want).mean() >= 0.85 and codes.min() >= 0 and codes.max() < c.codebook_size, n
        assert emu.qtts_encoder_encode(h, _ptr(x), 3, 331, _ptr(codes), None) != 0      # batch above max_batch
    finally:
        emu.hostemu_set_real_gemm(1 if FULL else 0)
        emu.qtts_encoder_destroy(h)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_speaker_orchestration_embedding_vs_oracle(emu, dtype):
    """qtts_speaker_* (speaker embedding): finalize() (DFT matrix, filterbank padding, conv tap layouts)
    + embed() on the emulated kernels against oracle/speaker_ref.py (torch.stft log-mel + the ECAPA-TDNN restatement that is pinned
    to the reference module): log-mel features and the embedding, for a length that is not a multiple of the hop."""
    import speaker_ref
    from qwen3_tts_amd.speaker import SpeakerEncoderConfig, fill_speaker_config, mel_filterbank_slaney
    emu.qtts_speaker_create.argtypes = [C.POINTER(_lib.SpeakerConfigC), C.POINTER(C.c_void_p)]
    emu.qtts_speaker_destroy.argtypes = [C.c_void_p]; emu.qtts_speaker_destroy.restype = None
    emu.qtts_speaker_bind.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    emu.qtts_speaker_finalize.argtypes = [C.c_void_p]
    emu.qtts_speaker_mel_frames.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    emu.qtts_speaker_embed.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    c = synth.speaker_small()
    w = {k: torch.from_numpy(v) for k, v in synth.speaker_weights(c).items()}
    cfg = SpeakerEncoderConfig.from_any(synth.cfg_dict(c))
    assert np.array_equal(mel_filterbank_slaney(24000, 1024, 128, 0, 12000), speaker_ref.mel_filterbank_slaney(24000, 1024, 128, 0, 12000))
    h = C.c_void_p()
    _ok(emu, emu.qtts_speaker_create(C.byref(fill_speaker_config(cfg, torch.float32 if dtype == "f32" else torch.bfloat16, 2, 8192)), C.byref(h)))
    try:
        for name, t in w.items():
            _lib.bind_tensor(emu.qtts_speaker_bind, h, name, t)
        _lib.bind_tensor(emu.qtts_speaker_bind, h, "mel_basis", torch.from_numpy(mel_filterbank_slaney(24000, 1024, 128, 0, 12000)))
        _ok(emu, emu.qtts_speaker_finalize(h))
        emu.hostemu_set_real_gemm(1)                                         # gemm_tap.hip itself (restored in `finally`)
        g = np.random.default_rng(9)
        for n in (4096, 6001):
            wav = (g.standard_normal((2, n)) * 0.2).clip(-1, 1).astype(np.float32)
            fr = C.c_int64()
            _ok(emu, emu.qtts_speaker_mel_frames(h, n, C.byref(fr)))
            with torch.no_grad():
                mel_ref = speaker_ref.mel_spectrogram(torch.from_numpy(wav)).transpose(1, 2)
                emb_ref = speaker_ref.speaker_encoder_forward(w, c, mel_ref).numpy()
            assert fr.value == mel_ref.shape[1]
            emb = np.zeros((2, c.enc_dim), np.float32)
            mels = np.zeros((2, fr.value, c.mel_dim), np.float32)
            _ok(emu, emu.qtts_speaker_embed(h, _ptr(wav), 2, n, _ptr(emb), _ptr(mels), None))
            if dtype == "f32":
                assert np.abs(mels - mel_ref.numpy()).max() <= 2e-4, n
                assert np.abs(emb - emb_ref).max() <= 1e-4 * max(1.0, float(np.abs(emb_ref).max())), n
            else:
                assert np.abs(mels - mel_ref.numpy()).max() <= 2e-4, n          # the mel front end stays fp32
                cos = [float(np.dot(a, b) / np.linalg.norm(a) / np.linalg.norm(b)) for a, b in zip(emb, emb_ref)]
                assert min(cos) >= 0.999, (n, cos)                               # bf16 ECAPA-TDNN GEMMs
        assert emu.qtts_speaker_embed(h, _ptr(wav), 3, n, _ptr(emb), None, None) != 0
    finally:
        emu.hostemu_set_real_gemm(1 if FULL else 0)
        emu.qtts_speaker_destroy(h)


def test_sampler_kernel_real_source_vs_hf_processors(emu):
    """sampling.hip's sample_kernel on the emulator against the oracle's restatement of the HF processor chain
    (RepetitionPenalty -> MinNewTokens -> Suppress -> Temperature -> TopK -> TopP, talker_ref.process_logits): the greedy
    pick is the argmax of the processed scores; under sampling every draw stays inside HF's support, the whole support is
    reached, and the empirical distribution over 3000 Philox (seed, step) pairs passes a chi-square test."""
    import talker_ref
    g = np.random.default_rng(77)
    for V, B in ((3072, 3), (2048, 2), (300, 2)):
        logits = (g.standard_normal((B, V + 4)) * 2.5).astype(np.float32)
        logits[0, 5] = logits[0, 9] = logits[0].max() + 1.0                   # a tie at the top: the lower index wins (torch.argmax)
        gen = g.integers(0, V, (B, 12)).astype(np.int32)
        n_gen = 7
        eos = V - 3
        logits[1, eos] = logits[1].max() + 3.0                                # EOS would win, but min_new_tokens blocks it
        sup = np.zeros(V, np.uint8)
        sup[V - 40:V - 8] = 1
        sup_list = [int(i) for i in np.nonzero(sup)[0]]
        lt, gt = torch.from_numpy(logits[:, :V].copy()), torch.from_numpy(gen[:, :n_gen].astype(np.int64))
        tok = np.zeros(B, np.int32)

        def launch(do_sample, top_k, top_p, temp, seed, step, min_new):
            rc = emu.hostemu_sample(_ptr(logits), V + 4, V, B, _ptr(gen), 12, n_gen, 1.3, eos, min_new, _ptr(sup), do_sample, top_k,
                                    top_p, temp, seed, 5, step, _ptr(tok))
            assert rc == 0, (emu.qtts_last_error() or b"").decode()
            return tok.copy()

        for min_new in (0, 9):
            want = talker_ref.process_logits(lt, gt, repetition_penalty=1.3, eos_id=eos, min_new_tokens=min_new, suppress=sup_list)
            assert np.array_equal(launch(0, 0, 1.0, 1.0, 0, 0, min_new), want.argmax(-1).numpy()), (V, min_new)
        # (0, 0.9) and (280, 0.75): the reference forwards ANY top_k / top_p to HF (IM:287-352) -- top-p with no top-k bound, and a
        # top-k beyond the 256-candidate fast path followed by top-p, both cut on the whole vocabulary (round 3)
        for top_k, top_p, temp in ((6, 1.0, 0.8), (50, 1.0, 0.9), (12, 0.7, 1.0), (0, 1.0, 1.3), (0, 0.9, 1.1), (280, 0.75, 1.0)):
            if (top_k == 0 or top_k > 256) and V > 300:
                continue                                                      # full-vocabulary multinomial: small case only
            sc = talker_ref.process_logits(lt, gt, repetition_penalty=1.3, eos_id=eos, min_new_tokens=9, suppress=sup_list,
                                           do_sample=True, temperature=temp, top_k=top_k, top_p=top_p)
            pr = torch.softmax(sc, -1).numpy()
            N = 3000 if V == 3072 else 1200
            counts = np.zeros_like(pr)
            for i in range(N):
                for b, tk in enumerate(launch(1, top_k, top_p, temp, 1000 + i // 7, i % 7, 9)):
                    counts[b, tk] += 1
            for b in range(B):
                assert (counts[b][pr[b] == 0] == 0).all(), "sampled a token outside HF's top-k / top-p support"
                supp = pr[b] > 0
                if top_k:
                    assert supp.sum() == top_k if top_p >= 1.0 else 1 <= supp.sum() <= top_k
                e, o = N * pr[b][supp], counts[b][supp]
                small = e < 5.0
                if small.sum() > 1:
                    e, o = np.append(e[~small], e[small].sum()), np.append(o[~small], o[small].sum())
                chi2, dof = float((((o - e) ** 2) / e).sum()), len(e) - 1
                assert chi2 < dof + 5.0 * np.sqrt(2.0 * max(dof, 1)) + 10.0, (V, top_k, top_p, b, chi2, dof)
        assert np.array_equal(launch(1, 50, 1.0, 0.9, 42, 3, 9), launch(1, 50, 1.0, 0.9, 42, 3, 9))       # same key -> same draw


def test_sampler_code_predictor_call_shape(emu):
    """The code predictor's sampler call (15 of a frame's 16 samplers: 2048 logits, top-k <= 64, no repetition penalty / suppress mask /
    EOS) through `sample_kernel_v2` against the oracle's HF processor chain: support exactly HF's top-k / top-p set, chi-square of the
    empirical distribution over Philox (seed, step) pairs, an all-equal row (every score a candidate: the general path) and a ragged
    leading dimension."""
    import talker_ref
    g = np.random.default_rng(123)
    V, B = 2048, 3
    logits = (g.standard_normal((B, V + 4)) * 2.5).astype(np.float32)
    logits[2, :V] = 0.25                                                      # all equal: top-k keeps every tied score (HF semantics)
    lt = torch.from_numpy(logits[:, :V].copy())
    tok = np.zeros(B, np.int32)

    def launch(top_k, top_p, temp, seed, step):
        rc = emu.hostemu_sample(_ptr(logits), V + 4, V, B, None, 0, 0, 1.0, -1, 0, None, 1, top_k, top_p, temp, seed, 3, step, _ptr(tok))
        assert rc == 0, (emu.qtts_last_error() or b"").decode()
        return tok.copy()

    for top_k, top_p, temp in ((50, 1.0, 0.9), (12, 0.7, 1.0)):
        sc = talker_ref.process_logits(lt, torch.zeros(B, 0, dtype=torch.long), do_sample=True, temperature=temp, top_k=top_k, top_p=top_p)
        pr = torch.softmax(sc, -1).numpy()
        N = 1500
        counts = np.zeros_like(pr)
        for i in range(N):
            for b, tk in enumerate(launch(top_k, top_p, temp, 500 + i // 5, i % 5)):
                counts[b, tk] += 1
        for b in range(B):
            if b == 2 and top_p < 1.0:
                # all-equal scores under top-p: WHICH of the tied tokens survive is decided by the sort order (torch.sort in HF, slot
                # order here) -- only the size of the surviving set is comparable
                ns, nd = int((pr[b] > 0).sum()), int((counts[b] > 0).sum())          # N uniform draws from ns tokens see ns (1 - e^(-N / ns)) of them
                assert 0.8 * ns * (1 - np.exp(-N / ns)) <= nd <= ns, (nd, ns)
                continue
            assert (counts[b][pr[b] == 0] == 0).all(), (top_k, top_p, b)
            supp = pr[b] > 0
            if b < 2:
                assert supp.sum() == top_k if top_p >= 1.0 else 1 <= supp.sum() <= top_k
            else:
                assert supp.sum() == V                                       # the tied row keeps everything
            e, o = N * pr[b][supp], counts[b][supp]
            small = e < 5.0
            if small.sum() > 1:
                e, o = np.append(e[~small], e[small].sum()), np.append(o[~small], o[small].sum())
            chi2, dof = float((((o - e) ** 2) / e).sum()), len(e) - 1
            assert chi2 < dof + 5.0 * np.sqrt(2.0 * max(dof, 1)) + 10.0, (top_k, top_p, b, chi2, dof)
    assert np.array_equal(launch(50, 1.0, 0.9, 42, 3), launch(50, 1.0, 0.9, 42, 3))


def _talker_emu(emu, t, w, max_batch, max_seq, dtype=None, use_graph=0):
    """Create + bind + finalize a talker handle on the emulation library (eager, or with the frame step captured and
    replayed through the emulated stream capture of hostemu/hip/hip_runtime.h)."""
    vp, i32 = C.c_void_p, C.c_int32
    emu.qtts_talker_create.argtypes = [C.POINTER(_lib.TalkerConfigC), C.POINTER(vp)]
    emu.qtts_talker_destroy.argtypes = [vp]; emu.qtts_talker_destroy.restype = None
    emu.qtts_talker_bind.argtypes = [vp, C.c_char_p, vp, i32, i32, C.POINTER(C.c_int64)]
    emu.qtts_talker_finalize.argtypes = [vp]
    emu.qtts_talker_prefill.argtypes = [vp, vp, i32, i32, C.POINTER(C.c_int32), vp, i32, vp, vp]
    emu.qtts_talker_generate.argtypes = [vp, C.POINTER(_lib.SamplingC), i32, i32, i32, C.POINTER(C.c_int32), i32, vp, vp, vp,
                                         C.POINTER(C.c_int32), vp]
    emu.qtts_talker_text_embed.argtypes = [vp, vp, i32, vp, vp]
    emu.qtts_talker_assemble_rows.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp]
    from qwen3_tts_amd.config import TalkerConfig
    c = TalkerConfig.from_any(synth.cfg_dict(t))
    tc = _lib.TalkerConfigC()
    for f in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads",
              "head_dim", "num_code_groups", "text_hidden_size", "codec_eos_token_id", "cp_vocab_size", "cp_hidden_size",
              "cp_intermediate_size", "cp_num_hidden_layers", "cp_num_attention_heads", "cp_num_key_value_heads", "cp_head_dim"):
        setattr(tc, f, int(getattr(c, f)))
    tc.rms_norm_eps, tc.rope_theta = float(c.rms_norm_eps), float(c.rope_theta)
    tc.cp_rms_norm_eps, tc.cp_rope_theta = float(c.cp_rms_norm_eps), float(c.cp_rope_theta)
    tc.weight_dtype, tc.max_batch, tc.max_seq, tc.use_graph = (_lib.QTTS_F32 if dtype is None else dtype), max_batch, max_seq, use_graph
    h = vp()
    _ok(emu, emu.qtts_talker_create(C.byref(tc), C.byref(h)))
    for name, x in w.items():
        _lib.bind_tensor(emu.qtts_talker_bind, h, name, x)
    _ok(emu, emu.qtts_talker_finalize(h))
    return h


def _talker_generate(emu, h, t, emb, mask, trailing, pad, max_new, eos=None, min_new=2):
    B, T, H = emb.shape
    n_pad = (mask == 0).sum(1).astype(np.int32)
    emb, trailing, pad = [np.ascontiguousarray(x, dtype=np.float32) for x in (emb, trailing, pad)]
    npad_c = (C.c_int32 * B)(*[int(x) for x in n_pad])
    _ok(emu, emu.qtts_talker_prefill(h, _ptr(emb), B, T, npad_c, _ptr(trailing), trailing.shape[1], _ptr(pad), None))
    sp = _lib.SamplingC()
    sp.do_sample, sp.subtalker_dosample, sp.repetition_penalty, sp.top_p, sp.subtalker_top_p = 0, 0, 1.05, 1.0, 1.0
    sp.temperature, sp.subtalker_temperature = 1.0, 1.0
    sup = [i for i in range(t.vocab_size - 1024, t.vocab_size) if i != t.codec_eos_token_id]
    sup_c = (C.c_int32 * len(sup))(*sup)
    mf = max(1, max_new - 1)
    codes = np.zeros((B, mf, t.num_code_groups), np.int64)
    hidden = np.zeros((B, mf, H), np.float32)
    tokens = np.full((B, max_new), -1, np.int64)
    nf = C.c_int32(0)
    _ok(emu, emu.qtts_talker_generate(h, C.byref(sp), max_new, min_new, t.codec_eos_token_id if eos is None else eos, sup_c, len(sup),
                                      _ptr(codes), _ptr(hidden), _ptr(tokens), C.byref(nf), None))
    n = int(nf.value)
    return codes[:, :n], tokens[:, :n + 1], hidden[:, :n]


@pytest.mark.parametrize("use_graph", [0, 1])
def test_talker_orchestration_greedy_vs_reference_golden(emu, golden_dir, use_graph):
    """The talker engine's real C++ -- weight packing (fused q|k|v, 16-row gate/up interleave, folded norm weights,
    streaming tile layout), prefill, the 15-pass code predictor + 28-layer-style frame step, EOS / finished-row
    bookkeeping, stop latch, eager and through the captured frame graph (graph cache keyed by everything a capture bakes
    in) -- on the emulated kernels, against the REFERENCE's greedy codes (tests/golden/talker_tiny.npz):
    bit-exact indices, final hidden state, and the early-EOS variant."""
    g = np.load(os.path.join(golden_dir, "talker_tiny.npz"))
    t = synth.talker_tiny()
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    h = _talker_emu(emu, t, w, max_batch=4, max_seq=64, use_graph=use_graph)
    try:
        args = [g[k] for k in ("embeds", "mask", "trailing", "tts_pad")]
        codes, tokens, hidden = _talker_generate(emu, h, t, *args, max_new=14)
        assert np.array_equal(tokens, g["tokens"]) and np.array_equal(codes, g["codes"])
        assert np.abs(hidden - g["hidden"]).max() <= 2e-3
        codes2, tokens2, _ = _talker_generate(emu, h, t, *args, max_new=14, eos=int(g["eos2"]))
        assert np.array_equal(tokens2, g["tokens_eos2"]) and np.array_equal(codes2, g["codes_eos2"])
        assert emu.qtts_talker_generate(h, None, 1, 1, 0, None, 0, None, None, None, None, None) != 0      # null arguments
    finally:
        emu.qtts_talker_destroy(h)


def test_talker_split_kv_attention_and_graph_switch(emu, golden_dir, qopt):
    """Long-sequence mode of the talker's decode attention: an engine told to use split-KV (two attention workgroups per
    (sequence, kv head) + merge kernel) from 20 keys on starts a generation on the short-sequence frame graph and SWITCHES to the
    long-sequence graph mid-way (the knobs default to max_seq > 512 / 320 keys; QTTS_ATTN_NSPLIT / QTTS_ATTN_SPLIT_FROM bring the
    switch into the range of the tiny golden).  The reference's greedy codes must still come out bit for bit."""
    qopt(emu, "QTTS_ATTN_NSPLIT", "2")
    qopt(emu, "QTTS_ATTN_SPLIT_FROM", "20")
    g = np.load(os.path.join(golden_dir, "talker_tiny.npz"))
    t = synth.talker_tiny()
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    h = _talker_emu(emu, t, w, max_batch=4, max_seq=64, use_graph=1)
    try:
        args = [g[k] for k in ("embeds", "mask", "trailing", "tts_pad")]
        assert args[0].shape[1] < 20 < args[0].shape[1] + 13, "the switch point must fall inside the generation"
        emu.qtts_talker_get_stats.argtypes = [C.c_void_p, C.POINTER(_lib.TalkerStatsC)]
        for _ in range(2):                      # second call: both graphs come from the cache
            codes, tokens, hidden = _talker_generate(emu, h, t, *args, max_new=14)
            assert np.array_equal(tokens, g["tokens"]) and np.array_equal(codes, g["codes"])
            assert np.abs(hidden - g["hidden"]).max() <= 2e-3
            st = _lib.TalkerStatsC()
            _ok(emu, emu.qtts_talker_get_stats(h, C.byref(st)))
            # the key range is partitioned by the LIVE length's bucket (512 keys = 2 x 256 here), not by max_seq; one long graph,
            # captured once and reused by the second call
            assert (st.long_graphs, st.attn_nsplit_last, st.attn_span_last) == (1, 2, 512)
    finally:
        emu.qtts_talker_destroy(h)


@pytest.mark.parametrize("use_graph", [0, 1] if FULL else [1])
def test_talker_stream_generation_equals_one_shot(emu, golden_dir, use_graph):
    """qtts_talker_stream_begin / _step / _end (resumable generation for streaming output): stepping the request in packets
    of 1, 3 or 5 frames yields, frame for frame, the codes of the one-shot generate -- i.e. the reference golden -- with a
    monotone `frames_total`, the stop latch reported once, for the normal and the early-EOS run; an abandoned stream
    reports the frames it produced.  Runs the captured-graph path (bursts of hipGraphLaunch between polls of the stop latch),
    which is what the GPU executes; the eager path too under QTTS_HOSTEMU_FULL=1."""
    vp, i32 = C.c_void_p, C.c_int32
    emu.qtts_talker_stream_begin.argtypes = [vp, C.POINTER(_lib.SamplingC), i32, i32, i32, C.POINTER(C.c_int32), i32, vp, vp, vp]
    emu.qtts_talker_stream_step.argtypes = [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp]
    emu.qtts_talker_stream_end.argtypes = [vp, vp, C.POINTER(C.c_int32), vp]
    g = np.load(os.path.join(golden_dir, "talker_tiny.npz"))
    t = synth.talker_tiny()
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    h = _talker_emu(emu, t, w, max_batch=4, max_seq=64, use_graph=use_graph)
    sup = [i for i in range(t.vocab_size - 1024, t.vocab_size) if i != t.codec_eos_token_id]
    sup_c = (C.c_int32 * len(sup))(*sup)

    def run(packet, eos, max_new=14, stop_after=None):
        emb, mask, trailing, pad = [np.ascontiguousarray(g[k]) for k in ("embeds", "mask", "trailing", "tts_pad")]
        B, T, H = emb.shape
        npad_c = (C.c_int32 * B)(*[int(x) for x in (mask == 0).sum(1)])
        _ok(emu, emu.qtts_talker_prefill(h, _ptr(emb), B, T, npad_c, _ptr(trailing), trailing.shape[1], _ptr(pad), None))
        sp = _lib.SamplingC()
        sp.do_sample, sp.subtalker_dosample, sp.repetition_penalty, sp.top_p, sp.subtalker_top_p = 0, 0, 1.05, 1.0, 1.0
        sp.temperature, sp.subtalker_temperature = 1.0, 1.0
        codes = np.zeros((B, max_new - 1, t.num_code_groups), np.int64)
        tokens = np.full((B, max_new), -1, np.int64)
        _ok(emu, emu.qtts_talker_stream_begin(h, C.byref(sp), max_new, 2, eos, sup_c, len(sup), _ptr(codes), None, None))
        total, fin, seen, packets = C.c_int32(0), C.c_int32(0), 0, 0
        while not fin.value:
            _ok(emu, emu.qtts_talker_stream_step(h, packet, C.byref(total), C.byref(fin), None))
            assert seen <= total.value <= seen + packet
            seen = total.value
            packets += 1
            if stop_after is not None and packets >= stop_after:
                break
        nf = C.c_int32(0)
        _ok(emu, emu.qtts_talker_stream_end(h, _ptr(tokens), C.byref(nf), None))
        assert nf.value == seen
        return codes[:, :seen], tokens[:, :seen + 1], packets

    try:
        for packet, early in ((1, False), (3, True), (5, False)) if not FULL else [(p, e) for p in (1, 3, 5) for e in (False, True)]:
            if not early:
                codes, tokens, _ = run(packet, t.codec_eos_token_id)
                assert np.array_equal(codes, g["codes"]) and np.array_equal(tokens, g["tokens"]), packet
            else:
                codes2, tokens2, _ = run(packet, int(g["eos2"]))
                assert np.array_equal(codes2, g["codes_eos2"]) and np.array_equal(tokens2, g["tokens_eos2"]), packet
        part, _, _ = run(2, t.codec_eos_token_id, stop_after=2)               # abandoned after two packets
        assert part.shape[1] == 4 and np.array_equal(part, g["codes"][:, :4])
        assert emu.qtts_talker_stream_step(h, 1, C.byref(C.c_int32()), C.byref(C.c_int32()), None) != 0    # no active stream
    finally:
        emu.qtts_talker_destroy(h)


def test_prompt_assembly_orchestration_vs_reference_golden(emu, golden_dir):
    """qtts_talker_text_embed + qtts_talker_assemble_rows through the real engine C++ (text-embedding table binding, the
    text_projection MLP on the tap GEMM, descriptor checks) on CPU kernels, driven by the host row plan, against what the
    REFERENCE's generate() hands to talker.generate in all five prompt modes (tests/golden/prompt_tiny.npz)."""
    from prompt_cases import CASES, load_case
    from qwen3_tts_amd.config import TalkerConfig
    from qwen3_tts_amd.model import build_prompt_plan, PLAN_PAD_ROW
    g = np.load(os.path.join(golden_dir, "prompt_tiny.npz"))
    t = synth.talker_tiny()
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t).items()}          # with the text embedding / projection
    cfg = TalkerConfig.from_any(synth.cfg_dict(t))
    h = _talker_emu(emu, t, w, max_batch=4, max_seq=64)
    H = t.hidden_size
    try:
        for name in CASES:
            c = load_case(g, name)
            plan = build_prompt_plan(cfg, c["ids"], c["languages"], c["speakers"], c["ins"], c["non_streaming_mode"], c["ref_ids"],
                                     c["voice_clone_prompt"])
            ids = np.ascontiguousarray(plan["text_ids"])
            proj = np.zeros((len(ids), H), np.float32)
            _ok(emu, emu.qtts_talker_text_embed(h, _ptr(ids), len(ids), _ptr(proj), None))
            spk = np.ascontiguousarray(np.stack([x.reshape(-1).numpy() for x in plan["spk_vectors"]]).astype(np.float32)) if plan["spk_vectors"] else None
            ref = np.ascontiguousarray(torch.cat(plan["ref_codes"], 0).numpy().astype(np.int64)) if plan["ref_codes"] else None
            desc = np.ascontiguousarray(plan["desc"])
            rows = np.zeros((desc.shape[0], H), np.float32)
            _ok(emu, emu.qtts_talker_assemble_rows(h, _ptr(desc), desc.shape[0], _ptr(proj), len(ids),
                                                   _ptr(spk) if spk is not None else None, 0 if spk is None else spk.shape[0],
                                                   _ptr(ref) if ref is not None else None, 0 if ref is None else ref.shape[0],
                                                   _ptr(rows), None))
            n, Tm, Tt = plan["n"], plan["Tm"], plan["Tt"]
            assert np.abs(rows[: n * Tm].reshape(n, Tm, H) - g[f"{name}_embeds"]).max() <= 2e-5, name
            assert np.abs(rows[n * Tm:].reshape(n, Tt, H) - g[f"{name}_trailing"]).max() <= 2e-5, name
            assert np.abs(proj[PLAN_PAD_ROW] - g[f"{name}_tts_pad"].reshape(-1)).max() <= 2e-5, name
        bad = np.array([10 ** 6], np.int64)                                          # a text id outside the table
        assert emu.qtts_talker_text_embed(h, _ptr(bad), 1, _ptr(np.zeros((1, H), np.float32)), None) != 0
    finally:
        emu.qtts_talker_destroy(h)


@pytest.mark.skipif(not FULL, reason="QTTS_HOSTEMU_FULL=1 (minutes on the emulator; the batch-32 case runs on the MI355X in tests/test_gpu_parity.py)")
def test_talker_orchestration_large_ragged_batch_vs_oracle(emu):
    """Twenty ragged, left-padded prompts (M = 20 rows per decode step, 40 in the code predictor's two-token first pass,
    several KV pages per sequence) through the real talker C++ on CPU kernels, against the oracle's greedy run."""
    import talker_ref
    t = synth.talker_tiny()
    wn = synth.talker_weights(t, with_text=False)
    w = {k: torch.from_numpy(v) for k, v in wn.items()}
    lens = [3 + (7 * i) % 13 for i in range(20)]
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(9), t, lens, 2, scale=0.5)
    sp = talker_ref.SamplingParams(do_sample=False, subtalker_dosample=False)
    with torch.no_grad():
        r = talker_ref.talker_generate(w, t, emb, mask, tr, pad, max_new_tokens=6, sp=sp)
    h = _talker_emu(emu, t, w, max_batch=20, max_seq=64)
    try:
        codes, tokens, _ = _talker_generate(emu, h, t, emb.numpy(), mask.numpy(), tr.numpy(), pad.numpy(), max_new=6)
        assert np.array_equal(tokens, r["tokens"].numpy()) and np.array_equal(codes, r["codes"].numpy())
    finally:
        emu.qtts_talker_destroy(h)
    # bf16 weights + bf16 KV pages: the product's serving configuration -- the LDS-staged skinny GEMM with two m-tiles
    # (M = 20 / 40), LDS-DMA staging of bf16 activations, bf16 attention pages.  No exact oracle; as in the GPU suite the
    # first frames must mostly agree with fp32, and the result must not depend on how the requests are batched
    # (B = 20 in one batch == two batches of 10 with the same left padding).
    h = _talker_emu(emu, t, w, max_batch=20, max_seq=64, dtype=_lib.QTTS_BF16)
    emu.hostemu_set_real_gemm(1)                                             # the bf16 prefill GEMMs: gemm_tap.hip itself
    try:
        c16, _, _ = _talker_generate(emu, h, t, emb.numpy(), mask.numpy(), tr.numpy(), pad.numpy(), max_new=3)
        assert float((c16[:, :2] == r["codes"].numpy()[:, :2]).mean()) >= 0.7
        lens2 = ([3 + (5 * i) % 11 for i in range(9)] + [15]) * 2
        e2, m2, tr2, pad2 = [x.numpy() for x in synth.rand_prompt(np.random.default_rng(10), t, lens2, 2, scale=0.5)]
        whole, _, _ = _talker_generate(emu, h, t, e2, m2, tr2, pad2, max_new=3)
        for half in (slice(0, 10), slice(10, 20)):
            part, _, _ = _talker_generate(emu, h, t, e2[half], m2[half], tr2[half], pad2, max_new=3)
            assert np.array_equal(part, whole[half])
    finally:
        emu.hostemu_set_real_gemm(1 if FULL else 0)
        emu.qtts_talker_destroy(h)


@pytest.mark.parametrize("cp_mlp", ["both", "1", "0"])
def test_talker_fp32_split_k_layer_chain_vs_oracle(emu, qopt, cp_mlp):
    """Round 5 (cp_mlp = "1"): the code predictor's passes >= 1 take the fp32 instantiation of the one-launch MLP (`cp_mlp_kernel<true, ...>`:
    the exact parity mode's fused leg; the o-projection before it no longer splits K, the talker stack keeps the split-K plan), and
    `cp_mlp_per_step` says so (QTTS_CP_MLP_F32=1: opt-in, the split-K plan is the faster one in fp32); cp_mlp = "both": q|k|v + attention +
    o-projection of those passes in `cp_attn_o_kernel<.., .., true>` as well (QTTS_CP_ATTN_O_F32=1; layer 0 without, layers >= 1 with the q|k|v
    front): the whole of a pass >= 1 on the fused construction, greedy fp32 against the oracle; cp_mlp = "0": round 4's plan on both stacks.
    Round 4: the ENGINE side of the fp32 split-K plan -- which GEMM of a layer splits, which one combines, which of the two
    residual buffers is current, the unsplit last layer writing where the caller reads -- at the real layer widths (hidden 1024,
    intermediate 3072, q width 2048: the 0.6B talker's and the code predictor's), which is where the plan engages; three layers per
    stack (first / middle / last take different branches), three code groups (pass 0 with two new tokens, pass 1 with its q|k|v row
    from the table).  Greedy fp32 against the oracle, and the same with QTTS_SKINNY8F_SPLITK=0 semantics checked by the oracle."""
    import talker_ref
    t = synth.TalkerCfg(vocab_size=1280, hidden_size=1024, intermediate_size=3072, num_hidden_layers=3, num_attention_heads=16,
                        num_key_value_heads=8, head_dim=128, num_code_groups=3, text_hidden_size=64, text_vocab_size=512,
                        cp_vocab_size=64, cp_hidden_size=1024, cp_intermediate_size=3072, cp_num_hidden_layers=3,
                        cp_num_attention_heads=16, cp_num_key_value_heads=8, cp_head_dim=128,
                        codec_eos_token_id=358, codec_think_id=362, codec_nothink_id=363, codec_think_bos_id=364, codec_think_eos_id=365,
                        codec_pad_id=356, codec_bos_id=357, im_start_token_id=500, im_end_token_id=501, tts_pad_token_id=502,
                        tts_bos_token_id=503, tts_eos_token_id=504, spk_id={"vivian": 1200}, spk_is_dialect={"vivian": False},
                        codec_language_id={"english": 301})
    wn = synth.talker_weights(t, with_text=False)
    w = {k: torch.from_numpy(v) for k, v in wn.items()}
    lens = [4, 2, 3]
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(19), t, lens, 2, scale=0.5)
    sp = talker_ref.SamplingParams(do_sample=False, subtalker_dosample=False)
    with torch.no_grad():
        r = talker_ref.talker_generate(w, t, emb, mask, tr, pad, max_new_tokens=4, sp=sp)
    qopt(emu, "QTTS_CP_MLP_F32", "0" if cp_mlp == "0" else "1")          # (fp32 engines take the fused launches on request only: talker_engine.hip finalize)
    qopt(emu, "QTTS_CP_ATTN_O_F32", "1" if cp_mlp == "both" else "0")
    h = _talker_emu(emu, t, w, max_batch=3, max_seq=32)
    try:
        codes, tokens, hidden = _talker_generate(emu, h, t, emb.numpy(), mask.numpy(), tr.numpy(), pad.numpy(), max_new=4)
        assert np.array_equal(tokens, r["tokens"].numpy()) and np.array_equal(codes, r["codes"].numpy())
        ref_h = r["hidden"].numpy()
        assert np.abs(hidden - ref_h).max() <= 2e-4 * max(1.0, float(np.abs(ref_h).max()))
        emu.qtts_talker_get_stats.argtypes = [C.c_void_p, C.POINTER(_lib.TalkerStatsC)]
        st = _lib.TalkerStatsC()
        _ok(emu, emu.qtts_talker_get_stats(h, C.byref(st)))
        want = (t.num_code_groups - 2) * t.cp_num_hidden_layers if cp_mlp != "0" else 0
        assert st.cp_mlp_per_step == want and st.cp_fused_per_step == (want if cp_mlp == "both" else 0) and st.cp_fused_giveups == 0, (st.cp_mlp_per_step, st.cp_fused_per_step)
        assert st.cp_fused_active == (1 if cp_mlp != "0" else 0) and st.cp_fused_capacity == (512 // 272 if cp_mlp != "0" else 0)
        # round 6: with both stages fused the whole layer is ONE launch (cp_layer_kernel<.., true, ...>: the merged construction's bit-exact leg)
        assert st.cp_layer_per_step == (want if cp_mlp == "both" else 0), st.cp_layer_per_step
    finally:
        emu.qtts_talker_destroy(h)


@pytest.mark.parametrize("cp_hidden", [256, 1024])
def test_talker_bf16_fused_attention_o_projection_in_the_frame_step(emu, qopt, cp_hidden):
    """Round 4: the ENGINE side of `cp_attn_o_kernel` -- the second packed copy of the o-projection (16-feature strips), the granule
    buffers and the frame serial the launch tags derive from, which passes take the fused launch (passes >= 1 of a bf16 engine at batch
    <= 8; pass 0 with its two new tokens keeps attn_cp0 + the decode GEMM) and which layers also take their q|k|v GEMM into it (layers
    >= 1 of a 1024-wide predictor: layer 0's row comes from the table) -- on a predictor with the real head geometry (16 query / 8 kv
    heads of 128) and a 256-wide (two 128-feature chunks; no q|k|v front) or 1024-wide (the real width; with the front) hidden state,
    greedy bf16, eager and through the captured frame graph: the codes and hidden states equal those of the same engine with
    QTTS_CP_ATTN_O=0 (separate launches) up to bf16 noise, the captured graph has (passes - 1) x layers fewer kernel nodes (the emulator
    runs a launch with the front as its two halves: one node less per fused layer either way), and a second generation on the same
    handle (granule buffers re-used, serial advanced) repeats the first bit for bit."""
    import dataclasses
    t = dataclasses.replace(synth.talker_tiny(), num_code_groups=4, cp_hidden_size=cp_hidden, cp_intermediate_size=256, cp_num_hidden_layers=2,
                            cp_num_attention_heads=16, cp_num_key_value_heads=8, cp_head_dim=128)
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(23), t, [5, 3, 6], 2, scale=0.5)
    args = (emb.numpy(), mask.numpy(), tr.numpy(), pad.numpy())
    emu.qtts_talker_get_stats.argtypes = [C.c_void_p, C.POINTER(_lib.TalkerStatsC)]
    emu.hostemu_set_real_gemm(1)
    res = {}
    try:
        for mode in ("1", "0"):
            qopt(emu, "QTTS_CP_ATTN_O", mode)
            for use_graph in ((0, 1) if cp_hidden == 256 else (1,)):     # (eager == graph: the 256-wide case; the 1024-wide one is 4x the emulation time)
                h = _talker_emu(emu, t, w, max_batch=4, max_seq=64, dtype=_lib.QTTS_BF16, use_graph=use_graph)
                try:
                    codes, tokens, hidden = _talker_generate(emu, h, t, *args, max_new=4)
                    if mode == "1":                                  # (granule buffers re-used, serial advanced)
                        codes2, tokens2, hidden2 = _talker_generate(emu, h, t, *args, max_new=4)
                        assert np.array_equal(codes, codes2) and np.array_equal(hidden, hidden2), (mode, use_graph)
                    st = _lib.TalkerStatsC()
                    _ok(emu, emu.qtts_talker_get_stats(h, C.byref(st)))
                    res[(mode, use_graph)] = (codes, hidden, int(st.graph_nodes))
                finally:
                    emu.qtts_talker_destroy(h)
    finally:
        emu.hostemu_set_real_gemm(1 if FULL else 0)
    for mode in ("1", "0"):                                          # eager == graph, as for every other path of the engine
        if (mode, 0) in res:
            assert np.array_equal(res[(mode, 0)][0], res[(mode, 1)][0]) and np.array_equal(res[(mode, 0)][1], res[(mode, 1)][1]), mode
    fused, plain = res[("1", 1)], res[("0", 1)]
    assert plain[2] - fused[2] == (t.num_code_groups - 2) * t.cp_num_hidden_layers, (plain[2], fused[2])
    n = min(fused[0].shape[1], plain[0].shape[1])
    assert n >= 2 and float((fused[0][:, :n] == plain[0][:, :n]).mean()) >= 0.9
    same = (fused[0][:, :n] == plain[0][:, :n]).all(axis=(0, 2))      # frames up to the first differing code see the same inputs
    k = int(np.argmin(same)) if not same.all() else n
    assert k >= 1
    assert np.abs(fused[1][:, :k] - plain[1][:, :k]).max() <= 2e-2 * max(1.0, float(np.abs(plain[1][:, :k]).max()))


def test_talker_bf16_fused_mlp_in_the_frame_step(emu, qopt):
    """Round 5: the ENGINE side of `cp_mlp_kernel` -- the gate|up operator packed by workgroup (XCD-major slices of the intermediate vector), the
    down operator in 16-feature strips, the two granule buffers, which launches take it (passes >= 1 of a bf16 engine at batch <= 8 that
    holds a place of its device's residency; pass 0 with its two new tokens keeps the two decode GEMMs) -- on a predictor with a 256-wide
    hidden state and a 1024-wide intermediate vector (16 features per workgroup), greedy bf16, eager and through the captured frame graph:
    codes and hidden states equal those of the same engine with QTTS_CP_MLP=0 up to bf16 noise, `cp_mlp_per_step` says which path ran, and a
    second generation on the same handle (granule buffers re-used, serial advanced) repeats the first bit for bit."""
    import dataclasses
    t = dataclasses.replace(synth.talker_tiny(), num_code_groups=4, cp_hidden_size=256, cp_intermediate_size=1024, cp_num_hidden_layers=2,
                            cp_num_attention_heads=16, cp_num_key_value_heads=8, cp_head_dim=128)
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(31), t, [5, 3, 6], 2, scale=0.5)
    args = (emb.numpy(), mask.numpy(), tr.numpy(), pad.numpy())
    emu.qtts_talker_get_stats.argtypes = [C.c_void_p, C.POINTER(_lib.TalkerStatsC)]
    per_step = (t.num_code_groups - 2) * t.cp_num_hidden_layers
    emu.hostemu_set_real_gemm(1)
    res = {}
    try:
        for mode in ("1", "0"):
            qopt(emu, "QTTS_CP_MLP", mode)
            for use_graph in (0, 1):
                h = _talker_emu(emu, t, w, max_batch=4, max_seq=64, dtype=_lib.QTTS_BF16, use_graph=use_graph)
                try:
                    codes, tokens, hidden = _talker_generate(emu, h, t, *args, max_new=4)
                    if mode == "1":
                        codes2, tokens2, hidden2 = _talker_generate(emu, h, t, *args, max_new=4)
                        assert np.array_equal(codes, codes2) and np.array_equal(hidden, hidden2), (mode, use_graph)
                    st = _lib.TalkerStatsC()
                    _ok(emu, emu.qtts_talker_get_stats(h, C.byref(st)))
                    assert st.cp_mlp_per_step == (per_step if mode == "1" else 0) and st.cp_fused_per_step == per_step and st.cp_fused_giveups == 0
                    res[(mode, use_graph)] = (codes, hidden)
                finally:
                    emu.qtts_talker_destroy(h)
    finally:
        emu.hostemu_set_real_gemm(1 if FULL else 0)
    for mode in ("1", "0"):
        assert np.array_equal(res[(mode, 0)][0], res[(mode, 1)][0]) and np.array_equal(res[(mode, 0)][1], res[(mode, 1)][1]), mode
    fused, plain = res[("1", 1)], res[("0", 1)]
    n = min(fused[0].shape[1], plain[0].shape[1])
    assert n >= 2 and float((fused[0][:, :n] == plain[0][:, :n]).mean()) >= 0.9
    same = (fused[0][:, :n] == plain[0][:, :n]).all(axis=(0, 2))
    k = int(np.argmin(same)) if not same.all() else n
    assert k >= 1
    assert np.abs(fused[1][:, :k] - plain[1][:, :k]).max() <= 2e-2 * max(1.0, float(np.abs(plain[1][:, :k]).max()))


@pytest.mark.parametrize("cp_hidden,cp_inter", [(256, 1024), (1024, 3072)])
def test_talker_bf16_whole_layer_as_one_launch_in_the_frame_step(emu, qopt, cp_hidden, cp_inter):
    """Round 6: the ENGINE side of `cp_layer_kernel` (csrc/cp_layer.hip) -- both fused stages of a code-predictor layer in ONE launch: the
    hidden rows between the o-projection's reducers and the MLP's phase A travel as tagged granules (a fifth granule buffer), the
    workgroup's gate|up block arrives by LDS-DMA, the MLP's reducer is the o-projection's (the residual stays in its registers).  Same
    arithmetic, same summation orders as the two launches: greedy bf16 frames AND hidden states must equal those of the same engine
    with QTTS_CP_LAYER=0 BIT FOR BIT (256-wide predictor: the q|k|v GEMM stays a launch of its own, eager and captured; 1024-wide = the
    released width: with the q|k|v front inside the launch, captured), `cp_layer_per_step` says which path ran, and a second generation
    on the same handle (five granule buffers re-used, serial advanced) repeats the first."""
    import dataclasses
    t = dataclasses.replace(synth.talker_tiny(), num_code_groups=4, cp_hidden_size=cp_hidden, cp_intermediate_size=cp_inter, cp_num_hidden_layers=2,
                            cp_num_attention_heads=16, cp_num_key_value_heads=8, cp_head_dim=128)
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(61), t, [5, 3, 6], 2, scale=0.5)
    args = (emb.numpy(), mask.numpy(), tr.numpy(), pad.numpy())
    emu.qtts_talker_get_stats.argtypes = [C.c_void_p, C.POINTER(_lib.TalkerStatsC)]
    per_step = (t.num_code_groups - 2) * t.cp_num_hidden_layers
    emu.hostemu_set_real_gemm(1)
    res = {}
    try:
        for mode in ("1", "0"):
            qopt(emu, "QTTS_CP_LAYER", mode)
            for use_graph in ((0, 1) if cp_hidden == 256 else (1,)):
                h = _talker_emu(emu, t, w, max_batch=4, max_seq=64, dtype=_lib.QTTS_BF16, use_graph=use_graph)
                try:
                    codes, tokens, hidden = _talker_generate(emu, h, t, *args, max_new=4)
                    if mode == "1":
                        codes2, tokens2, hidden2 = _talker_generate(emu, h, t, *args, max_new=4)
                        assert np.array_equal(codes, codes2) and np.array_equal(hidden, hidden2), (mode, use_graph)
                    st = _lib.TalkerStatsC()
                    _ok(emu, emu.qtts_talker_get_stats(h, C.byref(st)))
                    assert st.cp_layer_per_step == (per_step if mode == "1" else 0), (mode, st.cp_layer_per_step)
                    assert st.cp_mlp_per_step == per_step and st.cp_fused_per_step == per_step and st.cp_fused_giveups == 0
                    assert st.cp_fused_capacity == (1 if mode == "1" and cp_hidden == 1024 else 2)      # (layer engines may take half a compute unit's LDS together: one per device at the released dims -- 65 KB per workgroup)
                    res[(mode, use_graph)] = (codes, hidden)
                finally:
                    emu.qtts_talker_destroy(h)
    finally:
        emu.hostemu_set_real_gemm(1 if FULL else 0)
    ref = res[("0", 1)]
    assert ref[0].shape[1] >= 2
    for k, v in res.items():
        assert np.array_equal(v[0], ref[0]) and np.array_equal(v[1], ref[1]), f"QTTS_CP_LAYER={k[0]}, graph {k[1]}: differs from the two launches"


def test_talker_bf16_batch_above_16_splits_k_in_the_down_projections(emu, qopt):
    """Round 6 (BASELINE configs 4 and 5 run the frame step at batch 32): a bf16 engine at batch 17..32 sends the
    down-projections of the code predictor's passes >= 1 (K = 3072 at the released width) through `skinny2_ks_kernel` -- K split over the
    workgroups of a strip group, combined inside the launch.  `ks_split_per_step` says so (and 0 with QTTS_SKINNY_KS=0: skinny2_kernel); the
    split regroups fp32 sums, so hidden states agree to bf16-step accuracy and greedy codes almost everywhere; a second generation on the
    same handle (workspace re-used, serial advanced) repeats the first bit for bit; pass 0 (two tokens = 36 rows) keeps skinny2_kernel."""
    import dataclasses
    t = dataclasses.replace(synth.talker_tiny(), num_code_groups=3, cp_hidden_size=1024, cp_intermediate_size=3072, cp_num_hidden_layers=2,
                            cp_num_attention_heads=16, cp_num_key_value_heads=8, cp_head_dim=128)
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    emu.qtts_talker_get_stats.argtypes = [C.c_void_p, C.POINTER(_lib.TalkerStatsC)]
    emu.hostemu_set_real_gemm(1)
    res = {}
    try:
        for nb, want in ((18, (t.num_code_groups - 2) * t.cp_num_hidden_layers),):       # (M <= 16 never splits: skinny.hip ksplit_choice, kernel-level test)
            emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(63), t, [3 + i % 4 for i in range(nb)], 2, scale=0.5)
            args = (emb.numpy(), mask.numpy(), tr.numpy(), pad.numpy())
            for mode in ("1", "0"):
                qopt(emu, "QTTS_SKINNY_KS", mode)
                qopt(emu, "QTTS_SKINNY_KS_MINK", "3072")       # (the default floor, 6144, splits the talker's down-projection only: profiles/r06_skinny_ksplit.md)
                qopt(emu, "QTTS_CP_MLP32", "0")                # (the fused MLP launch at batch 9..32 takes these down-projections by default: its own test below)
                h = _talker_emu(emu, t, w, max_batch=32, max_seq=64, dtype=_lib.QTTS_BF16, use_graph=1)
                try:
                    codes, tokens, hidden = _talker_generate(emu, h, t, *args, max_new=3)
                    st = _lib.TalkerStatsC()
                    _ok(emu, emu.qtts_talker_get_stats(h, C.byref(st)))
                    assert st.ks_split_per_step == (want if mode == "1" else 0), (nb, mode, st.ks_split_per_step)
                    assert st.cp_fused_per_step == 0 and st.cp_fused_giveups == 0
                    if mode == "1" and want:
                        codes2, tokens2, hidden2 = _talker_generate(emu, h, t, *args, max_new=3)
                        assert np.array_equal(codes, codes2) and np.array_equal(hidden, hidden2)
                    res[(nb, mode)] = (codes, hidden)
                finally:
                    emu.qtts_talker_destroy(h)
    finally:
        emu.hostemu_set_real_gemm(1 if FULL else 0)
    (c1, h1), (c0, h0) = res[(18, "1")], res[(18, "0")]
    assert c1.shape == c0.shape and c1.shape[1] >= 2
    assert float(np.abs(h1 - h0).max()) <= 3e-2 * max(1.0, float(np.abs(h0).max())), float(np.abs(h1 - h0).max())
    assert float((c1 == c0).mean()) >= 0.95


def test_talker_bf16_batch_above_8_runs_the_code_predictors_mlp_as_one_launch(emu, qopt):
    """Round 6: a bf16 engine created for more than 8 rows runs the MLP of the code predictor's passes >= 1 as ONE launch
    at batch 9..32 too (`cp_mlp32_kernel`, released width 1024 / 3072, captured frame graph).  `cp_mlp_per_step` says which path ran (0 with
    QTTS_CP_MLP32=0: the two decode GEMMs), the engine holds a place of the device's account (208 registers: two such engines per device), a
    second generation on the same handle repeats the first bit for bit; fused and unfused add the same bf16 products in another fp32 order
    (kernel level: every 8-row block equals cp_mlp_kernel's bit for bit -- test_cp_mlp32_...), so hidden states agree to bf16-step accuracy
    and greedy codes almost everywhere; at batch <= 8 the same engine keeps skinny8 / the separate launches (cp_mlp_kernel is for engines
    created for <= 8 rows)."""
    import dataclasses
    t = dataclasses.replace(synth.talker_tiny(), num_code_groups=3, cp_hidden_size=1024, cp_intermediate_size=3072, cp_num_hidden_layers=2,
                            cp_num_attention_heads=16, cp_num_key_value_heads=8, cp_head_dim=128)
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    emu.qtts_talker_get_stats.argtypes = [C.c_void_p, C.POINTER(_lib.TalkerStatsC)]
    per_step = (t.num_code_groups - 2) * t.cp_num_hidden_layers
    emu.hostemu_set_real_gemm(1)
    res = {}
    try:
        for nb in (19,):
            emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(64), t, [3 + i % 4 for i in range(nb)], 2, scale=0.5)
            args = (emb.numpy(), mask.numpy(), tr.numpy(), pad.numpy())
            for mode in ("1", "0"):
                qopt(emu, "QTTS_CP_MLP32", mode)
                h = _talker_emu(emu, t, w, max_batch=32, max_seq=64, dtype=_lib.QTTS_BF16, use_graph=1)
                try:
                    codes, tokens, hidden = _talker_generate(emu, h, t, *args, max_new=3)
                    st = _lib.TalkerStatsC()
                    _ok(emu, emu.qtts_talker_get_stats(h, C.byref(st)))
                    assert st.cp_mlp_per_step == (per_step if mode == "1" else 0), (nb, mode, st.cp_mlp_per_step)
                    assert st.cp_fused_giveups == 0 and st.cp_layer_per_step == 0 and st.ks_split_per_step == 0
                    if mode == "1":
                        assert st.cp_fused_active == 1 and st.cp_fused_capacity == 2, (st.cp_fused_active, st.cp_fused_capacity)
                        codes2, tokens2, hidden2 = _talker_generate(emu, h, t, *args, max_new=3)
                        assert np.array_equal(codes, codes2) and np.array_equal(hidden, hidden2)
                    res[(nb, mode)] = (codes, hidden)
                finally:
                    emu.qtts_talker_destroy(h)
    finally:
        emu.hostemu_set_real_gemm(1 if FULL else 0)
    for nb in (19,):
        (c1, h1), (c0, h0) = res[(nb, "1")], res[(nb, "0")]
        assert c1.shape == c0.shape and c1.shape[1] >= 2
        assert float(np.abs(h1 - h0).max()) <= 3e-2 * max(1.0, float(np.abs(h0).max())), (nb, float(np.abs(h1 - h0).max()))
        assert float((c1 == c0).mean()) >= 0.95, nb


def test_fused_code_predictor_launch_is_admitted_per_device_by_residency(emu, qopt):
    """The fused code-predictor launch keeps all its workgroups resident, every one of which waits for others of the same launch.  The
    engine therefore keeps a per-device account of the register file at finalize (round 5: a compute unit's 512 registers per lane and SIMD
    against the shares of the fused engines' largest launches); an engine beyond that keeps the separate launches (more kernel nodes in its captured frame step), a place comes back when
    its engine is destroyed, a device that cannot hold one launch (a CPX partition: here occupancy 0) admits nobody -- and
    qtts_talker_stats says which path an engine took (cp_fused_per_step / cp_fused_launches_last / cp_fused_active / cp_fused_capacity)."""
    import dataclasses
    t = dataclasses.replace(synth.talker_tiny(), num_code_groups=4, cp_hidden_size=256, cp_intermediate_size=256, cp_num_hidden_layers=2,
                            cp_num_attention_heads=16, cp_num_key_value_heads=8, cp_head_dim=128)
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(29), t, [4, 3], 2, scale=0.5)
    args = (emb.numpy(), mask.numpy(), tr.numpy(), pad.numpy())
    emu.qtts_talker_get_stats.argtypes = [C.c_void_p, C.POINTER(_lib.TalkerStatsC)]
    fused_per_step = (t.num_code_groups - 2) * t.cp_num_hidden_layers

    def run(h):
        _talker_generate(emu, h, t, *args, max_new=3)
        st = _lib.TalkerStatsC()
        _ok(emu, emu.qtts_talker_get_stats(h, C.byref(st)))
        return st

    mk = lambda: _talker_emu(emu, t, w, max_batch=2, max_seq=32, dtype=_lib.QTTS_BF16, use_graph=1)
    # (1) the emulated device: 256 compute units of 512 registers per lane and SIMD; a code-predictor launch takes 184 of them -> 2 engines
    h = mk()
    try:
        st = run(h)
        assert st.cp_fused_capacity == 512 // 184 and st.cp_fused_active == 1 and st.cp_fused_giveups == 0
        assert st.cp_fused_per_step == fused_per_step and st.cp_fused_launches_last == fused_per_step * st.frames_run and st.frames_run >= 2
        n_fused = int(st.graph_nodes)
    finally:
        emu.qtts_talker_destroy(h)
    # (2) a cap of two places: the third engine alive at the same time keeps the separate launches; a destroyed engine's place is handed out again
    qopt(emu, "QTTS_CP_FUSED_MAX", "2")
    hs = [mk() for _ in range(3)]
    try:
        sts = [run(h) for h in hs]
        assert [s.cp_fused_active for s in sts] == [1, 1, 0] and [s.cp_fused_per_step for s in sts] == [fused_per_step, fused_per_step, 0]
        assert sts[2].cp_fused_launches_last == 0 and sts[2].cp_fused_capacity == 2
        assert sts[0].graph_nodes == n_fused and sts[2].graph_nodes - n_fused == fused_per_step, [s.graph_nodes for s in sts]
        emu.qtts_talker_destroy(hs.pop(0))
        hs.append(mk())
        st = run(hs[-1])
        assert st.cp_fused_active == 1 and st.graph_nodes == n_fused, "the place of a destroyed engine was not handed out again"
    finally:
        for h in hs:
            emu.qtts_talker_destroy(h)
    qopt(emu, "QTTS_CP_FUSED_MAX", None)
    # (3) a device that cannot keep one launch resident admits nobody
    qopt(emu, "QTTS_HOSTEMU_CPAO_BLOCKS_PER_CU", "0")
    h = mk()
    try:
        st = run(h)
        assert st.cp_fused_capacity == 0 and st.cp_fused_active == 0 and st.cp_fused_per_step == 0 and st.graph_nodes - n_fused == fused_per_step
    finally:
        emu.qtts_talker_destroy(h)


def test_code_predictor_with_more_than_five_layers_keeps_the_separate_launches(emu):
    """the fused launches tag their granules with slot = position * 5 + layer.  With a sixth layer, layer 5 of pass L would
    carry the tag of layer 0 of pass L + 1 in the same frame and the same buffers -- a stale granule would pass for a fresh one.  The
    reference's `cp_num_hidden_layers` is configurable (configuration_qwen3_tts.py:370-454; every released checkpoint has 5): an engine
    with six layers must run the separate launches -- stats say so -- and its greedy bf16 frames must be those of the same engine with the
    fused launches switched off by hand (i.e. nothing of the fused path ran), bit for bit."""
    import dataclasses
    t = dataclasses.replace(synth.talker_tiny(), num_code_groups=4, cp_hidden_size=256, cp_intermediate_size=1024, cp_num_hidden_layers=6,
                            cp_num_attention_heads=16, cp_num_key_value_heads=8, cp_head_dim=128)
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(41), t, [4, 3], 2, scale=0.5)
    args = (emb.numpy(), mask.numpy(), tr.numpy(), pad.numpy())
    emu.qtts_talker_get_stats.argtypes = [C.c_void_p, C.POINTER(_lib.TalkerStatsC)]
    emu.qtts_set_option.argtypes = [C.c_char_p, C.c_char_p]
    res = []
    for off in (False, True):
        if off:
            emu.qtts_set_option(b"QTTS_CP_ATTN_O", b"0"); emu.qtts_set_option(b"QTTS_CP_MLP", b"0")
        try:
            h = _talker_emu(emu, t, w, max_batch=2, max_seq=32, dtype=_lib.QTTS_BF16, use_graph=1)
            try:
                codes, tokens, hidden = _talker_generate(emu, h, t, *args, max_new=3)
                st = _lib.TalkerStatsC()
                _ok(emu, emu.qtts_talker_get_stats(h, C.byref(st)))
                assert st.cp_fused_per_step == 0 and st.cp_mlp_per_step == 0 and st.cp_fused_active == 0 and st.cp_fused_launches_last == 0
                res.append((codes, hidden, int(st.graph_nodes)))
            finally:
                emu.qtts_talker_destroy(h)
        finally:
            emu.qtts_set_option(b"QTTS_CP_ATTN_O", None); emu.qtts_set_option(b"QTTS_CP_MLP", None)
    assert res[0][2] == res[1][2] and np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_talker_orchestration_no_projection_vs_oracle(emu):
    """The 0.6B models' shape of the code predictor: talker hidden == predictor hidden, so small_to_mtp_projection is the
    identity (M:1171-1174) and the pass input row is the codec embedding itself.  Greedy fp32 against the oracle, then the bf16
    serving configuration (batch invariance; QTTS_PROBE_OUT2 dump for the build-variant comparison)."""
    import dataclasses
    import talker_ref
    t = dataclasses.replace(synth.talker_tiny(), hidden_size=128, intermediate_size=256)
    assert t.hidden_size == t.cp_hidden_size
    wn = synth.talker_weights(t, with_text=False)
    assert "code_predictor.small_to_mtp_projection.weight" not in wn
    w = {k: torch.from_numpy(v) for k, v in wn.items()}
    lens = [5, 9, 3]
    emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(19), t, lens, 2, scale=0.5)
    sp = talker_ref.SamplingParams(do_sample=False, subtalker_dosample=False)
    with torch.no_grad():
        r = talker_ref.talker_generate(w, t, emb, mask, tr, pad, max_new_tokens=5, sp=sp)
    h = _talker_emu(emu, t, w, max_batch=4, max_seq=64)
    try:
        codes, tokens, _ = _talker_generate(emu, h, t, emb.numpy(), mask.numpy(), tr.numpy(), pad.numpy(), max_new=5)
        assert np.array_equal(tokens, r["tokens"].numpy()) and np.array_equal(codes, r["codes"].numpy())
    finally:
        emu.qtts_talker_destroy(h)
    h = _talker_emu(emu, t, w, max_batch=4, max_seq=64, dtype=_lib.QTTS_BF16)
    emu.hostemu_set_real_gemm(1)
    try:
        c16, _, _ = _talker_generate(emu, h, t, emb.numpy(), mask.numpy(), tr.numpy(), pad.numpy(), max_new=4)
        n = min(c16.shape[1], r["codes"].shape[1], 2)
        assert float((c16[:, :n] == r["codes"].numpy()[:, :n]).mean()) >= 0.7
        one, _, _ = _talker_generate(emu, h, t, emb.numpy()[1:2], mask.numpy()[1:2], tr.numpy()[1:2], pad.numpy(), max_new=4)
        m = min(one.shape[1], c16.shape[1])
        assert np.array_equal(one[0, :m], c16[1, :m])
        if os.environ.get("QTTS_PROBE_OUT"):
            np.save(os.environ["QTTS_PROBE_OUT"].replace(".npy", "_noproj.npy"), c16)
    finally:
        emu.hostemu_set_real_gemm(1 if FULL else 0)
        emu.qtts_talker_destroy(h)


def test_talker_bf16_small_batch_staged_path(emu, golden_dir):
    """The bench configuration's code path at small batch (M = 3 rows per decode step, 6 in the code predictor's first pass):
    bf16 weights and KV, the LDS-staged single-m-tile skinny GEMM with LDS-DMA staging of bf16 activations, narrow strips.
    bf16 has no exact oracle: the first frames must mostly agree with the fp32 reference golden, and every row must come out
    the same whether it is generated in the batch or alone.  With QTTS_PROBE_OUT set the codes are also written there (the
    build-variant test compares them across builds: the same arithmetic must give the same bits)."""
    g = np.load(os.path.join(golden_dir, "talker_tiny.npz"))
    t = synth.talker_tiny()
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    h = _talker_emu(emu, t, w, max_batch=4, max_seq=64, dtype=_lib.QTTS_BF16)
    emu.hostemu_set_real_gemm(1)
    try:
        args = [g[k] for k in ("embeds", "mask", "trailing", "tts_pad")]
        codes, _, _ = _talker_generate(emu, h, t, *args, max_new=5)
        n = min(codes.shape[1], g["codes"].shape[1], 2)
        assert float((codes[:, :n] == g["codes"][:, :n]).mean()) >= 0.7
        for b in range(codes.shape[0]):
            one, _, _ = _talker_generate(emu, h, t, args[0][b:b + 1], args[1][b:b + 1], args[2][b:b + 1], args[3], max_new=5)
            m = min(one.shape[1], codes.shape[1])
            assert np.array_equal(one[0, :m], codes[b, :m]), b
        if os.environ.get("QTTS_PROBE_OUT"):
            np.save(os.environ["QTTS_PROBE_OUT"], codes)
    finally:
        emu.hostemu_set_real_gemm(1 if FULL else 0)
        emu.qtts_talker_destroy(h)


def test_talker_bf16_swiglu_single_strip_changes_nothing(emu, qopt):
    """Round 3: at batch <= 8 the frame step's gate|up GEMM runs from a second packed copy of the operator -- 8 gate + 8 up rows per
    strip, one strip per workgroup (ACT_SWIGLU8), twice the workgroups of the strip pairs.  Every output element is accumulated by the
    same MFMA sequence either way, so codes, tokens and hidden states must be bit-identical with QTTS_SWIGLU8=0 (strip pairs), at dims
    where the copy exists (hidden = 1024 | 2048 | 3072 | 6144, the K the batch <= 8 kernel is instantiated for)."""
    import dataclasses
    t = dataclasses.replace(synth.talker_tiny(), hidden_size=1024, intermediate_size=256, num_hidden_layers=1)     # (K = 1024: an instantiated K)
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    emb, mask, tr, pad = [x.numpy() for x in synth.rand_prompt(np.random.default_rng(8), t, [7, 5, 9], 2, scale=0.05)]
    outs = []
    for on in ("1", "0"):
        qopt(emu, "QTTS_SWIGLU8", on)
        h = _talker_emu(emu, t, w, max_batch=4, max_seq=32, dtype=_lib.QTTS_BF16)
        try:
            outs.append(_talker_generate(emu, h, t, emb, mask, tr, pad, max_new=4))
        finally:
            emu.qtts_talker_destroy(h)
    (c1, k1, h1), (c0, k0, h0) = outs
    assert c1.shape[1] >= 2 and np.array_equal(c1, c0) and np.array_equal(k1, k0) and np.array_equal(h1, h0)


def test_talker_bf16_prefill_bf16_handover_changes_nothing(emu, qopt):
    """Round 3: in bf16 mode the prefill's GEMM-only tensors (normed rows, attention output, SwiGLU product) are written as bf16 by
    their producers (rmsnorm16, attn_rows out16, the SwiGLU epilogue's C16) and read by the wide-K GEMM's bf16-activation
    instantiations.  The GEMM rounded them the same way while staging, so every code and every hidden state must come out
    bit-identical to the fp32 hand-over (QTTS_PREFILL_A16=0), at dims where the path is taken (hidden >= 512)."""
    import dataclasses
    t = dataclasses.replace(synth.talker_tiny(), hidden_size=512, intermediate_size=512, num_hidden_layers=1)
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    emb, mask, tr, pad = [x.numpy() for x in synth.rand_prompt(np.random.default_rng(5), t, [9, 6, 7], 2, scale=0.05)]
    outs = []
    for a16 in ("1", "0"):
        qopt(emu, "QTTS_PREFILL_A16", a16)
        h = _talker_emu(emu, t, w, max_batch=4, max_seq=32, dtype=_lib.QTTS_BF16)
        try:
            outs.append(_talker_generate(emu, h, t, emb, mask, tr, pad, max_new=3))
        finally:
            emu.qtts_talker_destroy(h)
    (c1, k1, h1), (c0, k0, h0) = outs
    assert c1.shape[1] >= 1 and np.array_equal(c1, c0) and np.array_equal(k1, k0) and np.array_equal(h1, h0)


@pytest.mark.skipif(not FULL, reason="QTTS_HOSTEMU_FULL=1 (2 min; a hardening pass of the emulator, not a parity gate)")
def test_results_do_not_depend_on_wave_scheduling_order(emu, codec, golden_dir):
    """The hardware runs the waves of a workgroup in no particular order; the emulator's default is ascending thread id.
    Re-run the kernel-level cases, the encoder, the speaker encoder, a short decode (whole and streamed) and a short talker
    generation with the fibers resumed
    in descending order and in a seeded shuffle of the waves (more under QTTS_HOSTEMU_FULL=1): a kernel that lacks a barrier between an LDS write and a
    read from another wave passes in one order and fails in another."""
    g = np.load(os.path.join(golden_dir, "talker_tiny.npz"))
    t = synth.talker_tiny()
    w = {k: torch.from_numpy(v) for k, v in synth.talker_weights(t, with_text=False).items()}
    try:
        for order in ((1, 2, 3, 4) if FULL else (1, 3)):
            emu.hostemu_set_fiber_order(order)
            for bf16 in (0, 1):
                test_gemm_tap_kernel_real_source(emu, bf16)
                test_skinny_kernel_real_source(emu, bf16)
            test_encoder_orchestration_codes_vs_reference_golden(emu, golden_dir, "f32")
            test_speaker_orchestration_embedding_vs_oracle(emu, "f32")
            c, cw, ch = codec                                                # decoder: short clip, one stream push sequence
            codes = np.random.default_rng(21).integers(0, c.codebook_size, (1, c.num_quantizers, 4))
            with torch.no_grad():
                ref = codec_ref.decoder_forward(cw, c, torch.from_numpy(codes))[:, 0].numpy()
            with real_gemm(emu):
                wav = np.zeros((1, 4 * c.total_upsample), np.float32)
                _ok(emu, emu.qtts_codec_forward(ch, _ptr(codes), 1, 4, _ptr(wav), None, None))
                assert np.sqrt(((wav - ref) ** 2).mean()) <= 1e-5, order
                _ok(emu, emu.qtts_codec_stream_begin(ch, 1))
                outs = []
                for a, b in ((0, 2), (2, 3), (3, 4)):
                    o = np.zeros((1, (b - a) * c.total_upsample), np.float32)
                    _ok(emu, emu.qtts_codec_stream_push(ch, _ptr(np.ascontiguousarray(codes[..., a:b])), b - a, _ptr(o), None))
                    outs.append(o)
                assert np.abs(np.concatenate(outs, axis=1) - ref).max() <= 5e-5, order
            h = _talker_emu(emu, t, w, max_batch=4, max_seq=64)
            try:
                codes, tokens, _ = _talker_generate(emu, h, t, *[g[k] for k in ("embeds", "mask", "trailing", "tts_pad")], max_new=4)
                assert np.array_equal(tokens, g["tokens"][:, :4]) and np.array_equal(codes, g["codes"][:, :3]), order
            finally:
                emu.qtts_talker_destroy(h)
    finally:
        emu.hostemu_set_fiber_order(0)


@pytest.mark.skipif(os.environ.get("QTTS_TEST_VARIANTS") != "1",
                    reason="the remaining A/B build variant (qwen3-tts_amd/build.py VARIANTS: attn_tail) on the emulator: one extra "
                           "emulator build, ~10 min -- enable with QTTS_TEST_VARIANTS=1")
def test_build_variants_agree_with_default_on_emulator(tmp_path):
    """attn_tail (KV beyond the 256-key window read 4 chunks per latency round), compiled into the emulated library with its -D
    flag (QTTS_HOSTEMU_DEFS): passes the decode-attention kernel test (long sequences included) and the talker goldens, and
    gives the same bf16 codes as the default build.  (The round-1 variants cp_pretable, cp_qkvtable, attn_cp and sampler_v2 were
    measured on hardware and are the default code now; the others were deleted -- profiles/r02_ab_variants.md.)"""
    import subprocess
    probes = {}
    for defs, sel in (("", "bf16_small_batch"),
                      ("-DQTTS_ATTN_TAIL_BATCH=1", "attn_decode or talker_orchestration_greedy or bf16_small_batch")):
        env = dict(os.environ, QTTS_HOSTEMU_DEFS=defs, QTTS_PROBE_OUT=str(tmp_path / f"probe{len(probes)}.npy"))
        env.pop("QTTS_TEST_VARIANTS", None)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k", sel], env=env,
                           capture_output=True, text=True)
        assert r.returncode == 0, (defs, r.stdout[-2000:])
        probes[defs] = np.load(env["QTTS_PROBE_OUT"])
    assert np.array_equal(probes[""], probes["-DQTTS_ATTN_TAIL_BATCH=1"])
