"""Summarise a rocprofv3 --pmc pass (rocpd sqlite): per-kernel average of one counter.
    python tools/rocpd_pmc.py gpurun_out/pmc1/pmc_results.db [--out profiles/x.md]"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
rows = cur.execute("select kernel_name, counter_name, value, grid_size_x, workgroup_size_x from counters_collection").fetchall()
agg = {}
for n, c, v, g, w in rows:
    k = (re.sub(r"\(.*", "", n).replace("qtts::", "")[:60], c)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += v
lines = ["| kernel | counter | dispatches | mean per dispatch | total |", "|---|---|---|---|---|"]
for (k, c), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:80]:
    lines.append(f"| `{k}` | {c} | {a[0]} | {a[1]/a[0]:.1f} | {a[1]:.0f} |")
out = "\n".join(lines); print(out)
if "--out" in sys.argv: open(sys.argv[sys.argv.index("--out") + 1], "w").write(out + "\n")
