"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / avg / total, plus inter-kernel gaps.
    python tools/rocpd_stats.py gpurun_out/prof1/perf_results.db [--out profiles/xxx.md]"""
import re, sqlite3, sys
db = sys.argv[1]
con = sqlite3.connect(db); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, start, end, grid_size_x, workgroup_size_x from kernels order by start").fetchall() if "grid_size_x" in cols else \
       cur.execute("select name, start, end, 0, 0 from kernels order by start").fetchall()
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
out = "\n".join(lines)
print(out)
if "--out" in sys.argv:
    open(sys.argv[sys.argv.index("--out") + 1], "w").write(out + "\n")
