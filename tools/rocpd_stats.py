"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / avg / total, plus inter-kernel gaps.
    python tools/rocpd_stats.py gpurun_out/prof1/perf_results.db [--out profiles/xxx.md]"""
import re, sqlite3, sys
db = sys.argv[1]
con = sqlite3.connect(db); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
# grid (work-items) and workgroup size in x: the column names differ between rocpd schema versions
gcol = next((c for c in ("grid_size_x", "grid_x", "grid_size") if c in cols), None)
wcol = next((c for c in ("workgroup_size_x", "workgroup_x", "workgroup_size") if c in cols), None)
if gcol and wcol:
    rows = cur.execute(f"select name, start, end, {gcol}, {wcol} from kernels order by start").fetchall()
else:
    print("rocpd_stats: no grid / workgroup columns in `kernels` (have: " + ", ".join(cols) + "): no per-shape table", file=sys.stderr)
    rows = cur.execute("select name, start, end, 0, 0 from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("qtts::", "")
    return n[:70]
agg = {}
for n, s, e, g, w in rows:
    k = short(n); a = agg.setdefault(k, [0, 0.0, 1e18, 0.0]); d = (e - s) / 1000.0
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    lines.append(f"| `{k}` | {a[0]} | {a[1]/1000:.3f} | {a[1]/a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100*a[1]/tot:.1f} |")
# gaps between consecutive kernels (same process), excluding gaps > 1 ms (phase changes)
gaps = [(rows[i+1][1] - rows[i][2]) / 1000.0 for i in range(len(rows) - 1)]
g = [x for x in gaps if 0 <= x < 1000]
import statistics
lines.append("")
lines.append(f"kernels: {len(rows)}, busy {tot/1000:.3f} ms; inter-kernel gaps (<1 ms): n={len(g)}, median {statistics.median(g):.2f} us, mean {sum(g)/len(g):.2f} us, total {sum(g)/1000:.3f} ms")
# the decode GEMM by shape (N, K), reconstructed from the instantiation and the grid: skinny8_kernel<SPW, FS, NP, NORM, NW> covers
# N = workgroups * FS * SPW output features and K = NP * NW * 64; bytes = N * K * 2 (bf16 weights, read once) -> fraction of the
# 8 TB/s HBM peak per class.  The same classes as bench.py's `roofline.classes` (measured live with per-launch events).
shape = {}
for n, s, e, g, w in rows:
    m = re.search(r"skinny8_kernel<(\d+), (\d+), (\d+), (true|false), (\d+)>", n)
    if not m or not w:
        continue
    spw, fs, np_, _, nw = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4), int(m.group(5))
    N, K = (g // w) * fs * spw, np_ * nw * 64
    a = shape.setdefault((N, K), [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1000.0
if shape:
    lines += ["", "| decode GEMM class (skinny8_kernel) | launches | avg us | MB (N*K*2) | GB/s | frac of 8 TB/s |", "|---|---|---|---|---|---|"]
    tb = tt = tn = 0
    for (N, K), a in sorted(shape.items(), key=lambda kv: -kv[1][1]):
        us = a[1] / a[0]; b = N * K * 2.0
        lines.append(f"| N={N} K={K} | {a[0]} | {us:.2f} | {b/1e6:.2f} | {b/us/1e3:.0f} | {b/us/1e3/8000:.4f} |")
        tb += b * a[0]; tt += a[1]; tn += a[0]
    lines.append(f"| all (launch-weighted) | {tn} | {tt/tn:.2f} | {tb/tn/1e6:.2f} | {tb/tt/1e3:.0f} | {tb/tt/1e3/8000:.4f} |")
out = "\n".join(lines)
print(out)
if "--out" in sys.argv:
    open(sys.argv[sys.argv.index("--out") + 1], "w").write(out + "\n")
