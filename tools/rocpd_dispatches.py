"""Per-dispatch listing of a rocprofv3 kernel trace (rocpd sqlite) for kernels whose name matches a pattern: grid, workgroup, LDS, registers, duration.
With a --pmc database: one row per (dispatch, counter).   python tools/rocpd_dispatches.py X.db gemm_dma [--pmc] [--out f.md]"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
pat = sys.argv[2]
def short(n): return re.sub(r"\(.*", "", n).replace("qtts::", "")[:48]
lines = []
if "--pmc" in sys.argv:
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    rows = cur.execute("select dispatch_id, kernel_name, counter_name, value, grid_size_x, workgroup_size_x, lds_block_size from counters_collection order by dispatch_id").fetchall() \
        if "lds_block_size" in cols else [r + (0,) for r in cur.execute("select dispatch_id, kernel_name, counter_name, value, grid_size_x, workgroup_size_x from counters_collection order by dispatch_id").fetchall()]
    disp = {}
    for d, n, c, v, g, w, l in rows:
        if not re.search(pat, n): continue
        e = disp.setdefault(d, dict(name=short(n), wgs=g // max(w, 1), lds=l)); e[c] = e.get(c, 0) + v
    names = sorted({k for e in disp.values() for k in e if k not in ("name", "wgs", "lds")})
    lines.append("| dispatch | kernel | workgroups | lds | " + " | ".join(names) + " |"); lines.append("|" + "---|" * (4 + len(names)))
    for d, e in sorted(disp.items()):
        lines.append(f"| {d} | `{e['name']}` | {e['wgs']} | {e['lds']} | " + " | ".join(f"{e.get(k, 0):.0f}" for k in names) + " |")
else:
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    want = [c for c in ("grid_size_x", "workgroup_size_x", "lds_size", "lds_block_size", "vgpr_count", "accum_vgpr_count", "sgpr_count", "scratch_size") if c in cols]
    rows = cur.execute(f"select name, start, end, {', '.join(want)} from kernels order by start").fetchall()
    lines.append("| # | kernel | us | " + " | ".join(want) + " |"); lines.append("|" + "---|" * (3 + len(want)))
    for i, r in enumerate(rows):
        if re.search(pat, r[0]): lines.append(f"| {i} | `{short(r[0])}` | {(r[2] - r[1]) / 1000:.2f} | " + " | ".join(str(x) for x in r[3:]) + " |")
    lines.append(""); lines.append("columns of `kernels`: " + ", ".join(cols))
out = "\n".join(lines); print(out)
if "--out" in sys.argv: open(sys.argv[sys.argv.index("--out") + 1], "w").write(out + "\n")
