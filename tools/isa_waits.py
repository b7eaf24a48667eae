#!/usr/bin/env python3
"""Scan the gfx950 code objects inside libqtts.so for SERIALIZED memory requests: a `s_waitcnt vmcnt(n)` with small n that is
followed by further global loads within a few instructions -- the signature of a conditional load whose result the compiler
merged with a previous register value (copy after the wait), or of a load behind a control-flow join with a pending load on
one side.  Round 2 found three hot kernels in that state (attn_tk: every K/V chunk request waited for all earlier ones).

    python tools/isa_waits.py [--lib qwen3-tts_amd/libqtts.so] [--window 40] [--max-n 1] [--filter attn]
"""
import argparse, os, re, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=os.path.join(ROOT, "qwen3-tts_amd", "libqtts.so"))
ap.add_argument("--window", type=int, default=40); ap.add_argument("--max-n", type=int, default=1)
ap.add_argument("--filter", default=""); ap.add_argument("--md", default=None)
a = ap.parse_args()
tmp = tempfile.mkdtemp()
try:
    local = os.path.join(tmp, "lib.so"); shutil.copy(a.lib, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, capture_output=True)
    rows = []
    for f in sorted(os.listdir(tmp)):
        if "gfx950" not in f: continue
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout
        name, body = None, []
        def flush():
            if not name or (a.filter and a.filter not in name): return
            ins = [l.split("//")[0].strip() for l in body if l.strip()]
            loads = [i for i, l in enumerate(ins) if l.startswith(("global_load", "buffer_load", "flat_load"))]
            hits = []
            for i, l in enumerate(ins):
                m = re.match(r"s_waitcnt vmcnt\((\d+)\)", l)
                if not m or int(m.group(1)) > a.max_n: continue
                before = sum(1 for j in loads if j < i)
                after = [j for j in loads if i < j <= i + a.window]
                if before and after: hits.append((i, int(m.group(1)), after[0] - i))
            if hits: rows.append((len(hits), name, len(ins), len(loads), hits[:6]))
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                flush(); name, body = m.group(1), []
            elif name is not None and line.startswith("\t"):
                body.append(line.split("\t", 1)[1] if "\t" in line else line)
        flush()
    rows.sort(reverse=True)
    out = ["| serialized waits | kernel | instructions | loads | first hits (instr index, vmcnt, distance to next load) |", "|---|---|---|---|---|"]
    for n, name, ni, nl, hits in rows:
        out.append(f"| {n} | `{name[:110]}` | {ni} | {nl} | {hits} |")
    txt = "\n".join(out)
    print(txt)
    if a.md: open(a.md, "w").write(txt + "\n")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
