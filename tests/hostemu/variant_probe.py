"""TEST INFRASTRUCTURE: dump the tokens sampling.hip draws on the emulator for a fixed set of logits / processor settings /
Philox keys (ties, an all-equal row that overflows the candidate buffer, repetition penalty, top-p, blocked EOS).  Run once
plain and once with QTTS_HOSTEMU_DEFS=-DQTTS_SAMPLER_V2=1: the two dumps must be identical (tests/test_hostemu.py,
QTTS_TEST_VARIANTS=1).  Usage: python tests/hostemu/variant_probe.py out.npy"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build as hb
lib = C.CDLL(hb.build())
vp, i32 = C.c_void_p, C.c_int32
lib.hostemu_sample.argtypes = [vp, i32, i32, i32, vp, i32, i32, C.c_float, i32, i32, vp, i32, i32, C.c_float, C.c_float, C.c_uint64, C.c_uint32, i32, vp]
g = np.random.default_rng(5)
res = []
P = lambda a: C.c_void_p(a.ctypes.data)
for V in (2048, 3072, 300, 4000):
    for top_k, top_p, temp, rep in ((50, 1.0, 0.9, 1.05), (50, 1.0, 0.9, 1.0), (12, 0.7, 1.0, 1.0), (64, 1.0, 1.3, 1.05), (1, 1.0, 0.9, 1.0)):
        B = 4
        logits = (g.standard_normal((B, V + 4)) * 3).astype(np.float32)
        logits[1, :V] = np.round(logits[1, :V])          # many exact ties
        logits[2, :V] = 0.25                              # all equal: > CAND_MAX ties -> general fallback
        gen = g.integers(0, V, (B, 12)).astype(np.int32)
        sup = (g.random(V) < 0.3).astype(np.uint8)
        eos = V - 1
        sup[eos] = 0
        for i in range(12):
            tok = np.full(B, -1, np.int32)
            rc = lib.hostemu_sample(P(logits), V + 4, V, B, P(gen) if rep != 1.0 else None, 12, 7, rep, eos, 9 if i % 2 else 0, P(sup), 1, top_k, top_p, temp, 1234 + i, 3, i, P(tok))
            assert rc == 0
            res.append(tok.copy())
np.save(sys.argv[1], np.stack(res))
print("dumped", len(res))
