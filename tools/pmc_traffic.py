#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the round's own rocprofv3 passes, stamped with a digest of the sources it describes.

    python tools/pmc_traffic.py --fetch-db <rocprofv3 --pmc FETCH_SIZE pass .db> [--trace-db <rocprofv3 --kernel-trace pass .db>] --model 1.7b

`traffic` of bench.py's roofline object = HBM bytes per decode-GEMM launch.  Method (MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE is
in KiB and under-reports by 2x on gfx950 -> bytes = value x 1024 x 2; counters in a pass of their own): over every `skinny8_kernel` dispatch of
the pass (410 of the frame step's 415 decode-GEMM launches; the other five share their instantiation with finalize-time work and are left out),
ratio = sum(FETCH_SIZE bytes) / sum(algorithmic bytes), where the algorithmic bytes of a dispatch are N x K x 2 with N and K reconstructed
from the instantiation and the grid (as tools/rocpd_stats.py does).  bench.py multiplies ITS algorithmic bytes per launch by that ratio.
With --trace-db the same classes' launch-weighted duration from the kernel trace gives `frac_rocprof` (the roofline fraction rocprofv3 sees).
The JSON carries a digest of csrc/skinny.hip + csrc/talker_engine.hip; bench.py reports `traffic: null` when the tree's digest differs.
The code predictor's fused launches (attention.hip: cp_attn_o_kernel, with / without the q|k|v front) ride along under "fused": dispatches,
FETCH_SIZE bytes and trace duration per launch -- bench.py prints them beside the decode GEMM's numbers (`roofline.fused_cp_launch`)."""
import argparse, hashlib, json, os, re, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_digest():
    h = hashlib.sha256()
    for f in ("skinny.hip", "talker_engine.hip", "attention.hip", "cp_mlp.hip", "cp_layer.hip", "cp_mlp32.hip"):
        with open(os.path.join(ROOT, "qwen3-tts_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def shape_of(name, grid, wg):
    m = re.search(r"skinny8_kernel<(\d+), (\d+), (\d+), (true|false), (\d+)>", name)
    if not m or not wg:
        return None
    spw, fs, np_, nw = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(5))
    return (grid // wg) * fs * spw, np_ * nw * 64


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch-db", required=True)
    ap.add_argument("--trace-db")
    ap.add_argument("--model", default="1.7b")
    ap.add_argument("--source", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    a = ap.parse_args()
    cur = sqlite3.connect(a.fetch_db).cursor()
    rows = cur.execute("select kernel_name, counter_name, value, grid_size_x, workgroup_size_x from counters_collection").fetchall()
    fetch = alg = 0.0
    n = 0
    fused = {}                                   # "front" | "attn_o" -> [dispatches, fetch bytes, trace launches, trace us]
    def fused_key(name):
        if "cp_mlp_kernel<" in name:
            return "mlp"
        m = re.search(r"cp_layer_kernel<(true|false), ", name)          # (round 6: <QKV, F32, ...> -- the whole layer in one launch)
        if m:
            return "layer_front" if m.group(1) == "true" else "layer"
        m = re.search(r"cp_attn_o_kernel<(true|false), (true|false)[,>]", name)
        return None if not m else ("front" if m.group(2) == "true" else "attn_o")
    for name, cname, v, g, w in rows:
        if cname != "FETCH_SIZE":
            continue
        fk = fused_key(name)
        if fk:
            f = fused.setdefault(fk, [0, 0.0, 0, 0.0]); f[0] += 1; f[1] += v * 1024.0 * 2.0
            continue
        s = shape_of(name, g, w)
        if s is None:
            continue
        fetch += v * 1024.0 * 2.0
        alg += s[0] * s[1] * 2.0
        n += 1
    assert n > 0, "no skinny8_kernel dispatches with FETCH_SIZE in this pass"
    rec = {"ratio_traffic_over_algorithmic": round(fetch / alg, 4), "dispatches": n, "bytes_per_launch_skinny8": round(fetch / n),
           "algorithmic_bytes_per_launch_skinny8": round(alg / n), "source": a.source or os.path.basename(a.fetch_db)}
    if a.trace_db:
        c2 = sqlite3.connect(a.trace_db).cursor()
        cols = [r[1] for r in c2.execute("pragma table_info(kernels)")]
        gcol = next(c for c in ("grid_size_x", "grid_x", "grid_size") if c in cols)
        wcol = next(c for c in ("workgroup_size_x", "workgroup_x", "workgroup_size") if c in cols)
        tb = tt = tn = 0.0
        for name, s0, e0, g, w in c2.execute(f"select name, start, end, {gcol}, {wcol} from kernels"):
            fk = fused_key(name)
            if fk:
                f = fused.setdefault(fk, [0, 0.0, 0, 0.0]); f[2] += 1; f[3] += (e0 - s0) / 1000.0
                continue
            s = shape_of(name, g, w)
            if s is None:
                continue
            tb += s[0] * s[1] * 2.0; tt += (e0 - s0) / 1000.0; tn += 1
        rec["rocprof_avg_launch_us"] = round(tt / tn, 3)
        rec["frac_rocprof"] = round(tb / tt / 1e3 / 8000.0, 4)
        rec["rocprof_launches"] = int(tn)
    if fused:
        rec["fused"] = {k: {"dispatches": f[0], "fetch_bytes_per_launch": round(f[1] / f[0]) if f[0] else None, "rocprof_launches": f[2],
                            "rocprof_avg_launch_us": round(f[3] / f[2], 3) if f[2] else None} for k, f in fused.items()}
    out = {"_doc": __doc__.split("\n\n")[1].replace("\n", " "), "kernel_digest": kernel_digest(), a.model: rec}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out[a.model]))
