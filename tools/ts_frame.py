"""In-kernel phase timestamps of the frame step (measuring tool, not a test, not the bench).

Runs the talker at real dims on the `tstamp` build variant (csrc/tstamp.h: wave 0 of the first and of the last workgroup of
every decode-GEMM / decode-attention / sampler launch reads the 100 MHz constant clock at up to 6 points), reads the three
per-unit logs back, orders all records by entry time and prints, per kernel class,
  * the phase averages inside the kernel (us from kernel entry), and
  * the boundary: entry of this launch minus the LAST stamp of the launch before it in the timeline (graph node to graph node).

    python qwen3-tts_amd/build.py --variant tstamp
    python tools/ts_frame.py --model 1.7b --frames 12 [--no-graph] [--json out.json]
"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "qwen3-tts_amd", "libqtts_tstamp.so")
os.environ["QTTS_LIBRARY"] = LIB
import numpy as np, torch
import synth
from qwen3_tts_amd.talker import TalkerEngine

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="1.7b"); ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--prompt", type=int, default=0, help="extra prompt tokens (longer KV)")
ap.add_argument("--json", default=None)
a = ap.parse_args()

REC = np.dtype([("t", "<u8", 6), ("kind", "<i4"), ("a", "<i4"), ("b", "<i4"), ("blk", "<i4")])
lib = C.CDLL(LIB)
CAP = 1 << 15


def drain():
    out = []
    for unit in ("skinny", "attn", "sample", "cpmlp", "cplayer"):
        fn = getattr(lib, f"qtts_debug_tslog_{unit}")
        fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
        buf = np.zeros(CAP, dtype=REC)
        n = fn(buf.ctypes.data, CAP)
        if n < 0: raise RuntimeError(f"qtts_debug_tslog_{unit} failed")
        out.append(buf[:n].copy())
    return np.concatenate(out)


def cheap(shapes, std_of):
    base = np.random.default_rng(0).standard_normal(1 << 20, dtype=np.float32)
    out = {}
    for k, shp in shapes.items():
        v = np.resize(base, int(np.prod(shp))).reshape(shp) * np.float32(std_of(k, shp))
        if "norm" in k and k.endswith("weight"): v = v * 0 + 1
        if "head" in k:      # the tiled base repeats every 2^20 values: identical head rows = tied logits, which the sampler sees
            v = np.random.default_rng(__import__("zlib").crc32(k.encode())).standard_normal(shp, dtype=np.float32) * np.float32(std_of(k, shp))
        out[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


B, F = a.batch, a.frames
t = {"1.7b": synth.talker_17b, "0.6b": synth.talker_06b}[a.model]()
w = cheap(synth.talker_param_shapes(t, with_text=False), lambda k, s: 0.08 if ("head" in k) else 0.02)
eng = TalkerEngine(t, w, weight_dtype=torch.bfloat16, max_batch=B, max_seq=64 + a.prompt + F + 8, use_graph=not a.no_graph)
del w
lens = [24 + 4 * (i % 8) + 12 + a.prompt for i in range(B)]
emb, mask, tr, pad = synth.rand_prompt(np.random.default_rng(1), t, lens, 1)
sup = [i for i in range(t.vocab_size - 1024, t.vocab_size) if i != t.codec_eos_token_id]
kw = dict(max_new_tokens=F + 1, min_new_tokens=F + 1, suppress_tokens=sup, output_hidden_states=False)
eng.generate(emb, mask, tr, pad, seed=0, **kw); torch.cuda.synchronize()      # warm-up (graph capture, caches)
drain()
eng.generate(emb, mask, tr, pad, seed=1, **kw); torch.cuda.synchronize()
r = drain()
print(f"{len(r)} records ({'eager' if a.no_graph else 'hipGraph'} launches, batch {B}, {F} frames)")

TICK_US = 0.01                                  # s_memrealtime: 100 MHz
order = np.argsort(r["t"][:, 0], kind="stable")
r = r[order]
KIND = {0: "decode GEMM", 1: "attn_cp", 2: "attn_tk", 3: "sampler", 4: "cp_attn_o", 5: "cp_mlp", 6: "cp_layer", 7: "split-K GEMM"}
PHASES = {0: ["issued", "arrived", "mfma+lds", "barrier", "stored"],
          1: ["issued", "arrived", "normed", "barrier", "stored"],
          2: ["arrived", "normed", "keys folded", "barrier", "stored"],
          3: ["arrived", "bound", "ranked", "drawn", "rows out"],
          4: ["arrived", "attended", "published", "ticket", "reduced"],
          7: ["issued", "arrived", "mfma+lds", "parts out|in", "stored"],      # (round 6: skinny2_ks_kernel at batch 17..32; blk bit 16 = the strip group's reducer)
          6: ["o part out", "hidden in", "act out", "part out", "reduced"],      # (round 6: the whole layer in one launch; blk bit 16 = a reducer)
          5: ["A mfma", "act out", "slice in", "part out", "reduced"]}      # (round 5: the fused MLP launch; blk bit 16 = a reducer of XCD 7)      # (round 4: attention + o-projection in one launch; blk bit 16 = the chunk's last arriver)
# boundary: entry of a launch minus the latest stamp of any record of the launch before it.  Two records (first / last
# workgroup) of one launch share (kind, a, b) and lie within 2 us of each other.
ends = r["t"].max(axis=1)
gap = np.full(len(r), np.nan)
launch_id = np.zeros(len(r), dtype=np.int64)
lid = 0
for i in range(1, len(r)):
    same = (r["kind"][i] == r["kind"][i - 1] and r["a"][i] == r["a"][i - 1] and r["b"][i] == r["b"][i - 1]
            and r["blk"][i] != r["blk"][i - 1] and (r["t"][i, 0] - r["t"][i - 1, 0]) * TICK_US < 2.0)
    if not same: lid += 1
    launch_id[i] = lid
n_l = lid + 1
l_entry = np.array([r["t"][launch_id == k, 0].min() for k in range(n_l)], dtype=np.float64)
l_end = np.array([ends[launch_id == k].max() for k in range(n_l)], dtype=np.float64)
l_gap = np.full(n_l, np.nan); l_gap[1:] = (l_entry[1:] - l_end[:-1]) * TICK_US
l_pitch = np.full(n_l, np.nan); l_pitch[1:] = (l_entry[1:] - l_entry[:-1]) * TICK_US
for i in range(len(r)): gap[i] = l_gap[launch_id[i]]

rows = []
keys = sorted({(int(k), int(x), int(y)) for k, x, y in zip(r["kind"], r["a"], r["b"]) if k not in (1, 2, 4, 5, 6, 7)})
ks_keys = sorted({(int(x), int(y)) for k, x, y in zip(r["kind"], r["a"], r["b"]) if k == 7})
keys += [(1, -1, 0), (2, -1, 0), (4, -1, 0), (4, -1, 1), (5, -1, 0), (5, -1, 1), (6, -1, 0), (6, -1, 1)]        # the attentions: all cache lengths together; cp_attn_o: other workgroups | last arrivers
keys += [(7, x, y, red) for (x, y) in ks_keys for red in (0, 1)]
for key in keys:
    k = key[0]
    sel = (r["kind"] == k) if key[1] < 0 else ((r["kind"] == k) & (r["a"] == key[1]) & (r["b"] == key[2]))
    if k in (4, 5, 6): sel = sel & ((r["blk"] >> 16) == key[2])
    if k == 7: sel = sel & ((r["blk"] >> 16) == key[3])
    if not sel.any(): continue
    rr = r[sel]
    ph = (rr["t"][:, 1:].astype(np.float64) - rr["t"][:, :1].astype(np.float64)) * TICK_US
    ph[rr["t"][:, 1:] == 0] = np.nan
    g = gap[sel]; g = g[np.isfinite(g) & (g < 20)]
    name = KIND[k] + (f" K={key[1]} N={key[2]}" + (" (reducer)" if key[3] else " (producer)") if k == 7 else f" K={key[1]} N={key[2]}" if k == 0 else (f" V={key[1]}" if k == 3 else ((" (last arriver)" if key[2] else " (others)") if k == 4 else ((" (reducer, XCD 7)" if key[2] else " (others)") if k == 5 else ((" (reducer)" if key[2] else " (others)") if k == 6 else "")))))
    rows.append(dict(kernel=name, records=int(sel.sum()), phases_us={n: round(float(np.nanmean(ph[:, j])), 2) for j, n in enumerate(PHASES[k])},
                     in_kernel_us=round(float(np.nanmean(np.nanmax(ph, axis=1))), 2),
                     boundary_us_median=round(float(np.median(g)), 2) if len(g) else None))
print(f"{'kernel':34s} {'n':>5s}  " + "  ".join(f"{'phase'+str(j+1):>11s}" for j in range(5)) + f"  {'in-kernel':>9s} {'boundary':>8s}")
for x in rows:
    print(f"{x['kernel']:34s} {x['records']:5d}  " + "  ".join(f"{n[:6]:>6s}{v:5.2f}" for n, v in x["phases_us"].items())
          + f"  {x['in_kernel_us']:9.2f} {x['boundary_us_median'] if x['boundary_us_median'] is not None else float('nan'):8.2f}")
l_kind = np.array([r["kind"][launch_id == k][0] for k in range(n_l)])
if (l_kind == 4).any():       # the fused launch ends with its slowest reducer: the launch's length, not the mean of its workgroups, is what the next node waits for
    ln = (l_end - l_entry)[l_kind == 4] * TICK_US
    print(f"cp_attn_o launch length (first logged entry -> last stamp of any logged workgroup): mean {ln.mean():.2f} us, median {np.median(ln):.2f}, "
          f"p90 {np.percentile(ln, 90):.2f}, max {ln.max():.2f}, n={len(ln)}")
if (l_kind == 5).any():
    ln = (l_end - l_entry)[l_kind == 5] * TICK_US
    print(f"cp_mlp launch length (first logged entry -> last stamp of any logged workgroup): mean {ln.mean():.2f} us, median {np.median(ln):.2f}, "
          f"p90 {np.percentile(ln, 90):.2f}, max {ln.max():.2f}, n={len(ln)}")
if (l_kind == 6).any():
    ln = (l_end - l_entry)[l_kind == 6] * TICK_US
    print(f"cp_layer launch length (first logged entry -> last stamp of any logged workgroup): mean {ln.mean():.2f} us, median {np.median(ln):.2f}, "
          f"p90 {np.percentile(ln, 90):.2f}, max {ln.max():.2f}, n={len(ln)}")
fin = l_pitch[np.isfinite(l_pitch) & (l_pitch < 30)]
print(f"launch pitch (entry to entry, instrumented launches that follow one another): median {np.median(fin):.2f} us, mean {fin.mean():.2f} us, n={len(fin)}")
if a.json:
    json.dump(dict(mode="eager" if a.no_graph else "graph", batch=B, frames=F, model=a.model, rows=rows,
                   pitch_us_median=float(np.median(fin))), open(a.json, "w"), indent=1)
