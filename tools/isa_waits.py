#!/usr/bin/env python3
"""Scan the gfx950 code objects inside libqtts.so for SERIALIZED memory requests: a `s_waitcnt vmcnt(n)` with small n that is
followed by further global loads within a few instructions -- the signature of a conditional load whose result the compiler
merged with a previous register value (copy after the wait), or of a load behind a control-flow join with a pending load on
one side.  Round 2 found three hot kernels in that state (attn_tk: every K/V chunk request waited for all earlier ones).

    python tools/isa_waits.py [--lib qwen3-tts_amd/libqtts.so] [--window 40] [--max-n 1] [--filter attn]

`kernels(lib)` / `scan(lib, ...)` are importable (tests/test_host_logic.py pins the request bursts of the frame step's kernels).
"""
import argparse, os, re, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    """{demangled kernel name: [instruction text, ...]} of every gfx950 code object embedded in `lib`."""
    tmp = tempfile.mkdtemp()
    out = {}
    try:
        local = os.path.join(tmp, "lib.so"); shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f: continue
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", os.path.join(tmp, f)], capture_output=True,
                                 text=True, check=True).stdout
            name = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if m:
                    name = m.group(1); out[name] = []
                elif name is not None and line.startswith("\t"):
                    ins = line.split("\t", 1)[1].split("//")[0].strip()
                    if ins: out[name].append(ins)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def dma_barriers(lib):
    """Kernels that stage tiles by LDS-DMA (`global_load_lds` / `buffer_load ... lds`): {kernel: (barriers, barriers reached in
    program-text order after a DMA request without a `s_waitcnt vmcnt(0)` in between)}.  The second number must be 0: a barrier
    orders the WAVES, the DMA data is in LDS only once the requesting wave's vmcnt has drained."""
    out = {}
    for name, ins in kernels(lib).items():
        is_dma = lambda i: "global_load_lds" in i or (i.startswith("buffer_load") and i.rstrip().endswith(" lds"))
        if not any(is_dma(i) for i in ins):
            continue
        if "gemm_ring_kernel" in name:      # counted waits by design: ring_barriers() below
            continue
        tot = bad = 0
        waited = True
        for i in ins:
            if is_dma(i):
                waited = False
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", i)
            if m and int(m.group(1)) == 0:
                waited = True
            if i.startswith("s_barrier"):
                tot += 1
                bad += 0 if waited else 1
        out[name] = (tot, bad)
    return out


def ring_barriers(lib):
    """gemm_ring_kernel<NST, AH, RING_A> (round 6) keeps LDS-DMA requests in flight across its barriers under COUNTED waits: {kernel: (barriers,
    the vmcnt immediates found in the 24 instructions before each barrier, barriers with a `s_waitcnt lgkmcnt(0)` among them)}.  What the source
    promises and the test pins: every barrier has a vmcnt wait right before it; the two barriers of the steady-state loop wait with exactly
    LPS * (NST - 2) (LPS = 2 requests per wave and step, 4 when the A tile rides in the ring); no wait of the kernel allows more than the
    prologue's LPS * (NST - 1); every barrier is preceded by lgkmcnt(0) (the fragment reads of the buffer about to be re-requested are in registers)."""
    out = {}
    for name, ins in kernels(lib).items():
        if "gemm_ring_kernel" not in name: continue
        bars = [i for i, x in enumerate(ins) if x.startswith("s_barrier")]
        vm, lg = [], 0
        for b in bars:
            win = ins[max(0, b - 24):b]
            vm.append([int(m.group(1)) for x in win for m in [re.search(r"s_waitcnt.*vmcnt\((\d+)\)", x)] if m])
            lg += any(re.search(r"s_waitcnt.*lgkmcnt\(0\)", x) for x in win)
        allv = [int(m.group(1)) for x in ins for m in [re.search(r"s_waitcnt.*vmcnt\((\d+)\)", x)] if m]
        out[name] = (len(bars), vm, lg, max(allv) if allv else -1)
    return out


def polling_reloads(lib, name_filter=""):
    """Kernels that poll granules with `sc1` buffer loads (granule.h): {kernel: (sc1 loads, sc1 loads that sit behind a backward branch target,
    i.e. inside a loop)}.  A raw buffer load is a read-only intrinsic: in a loop without a store or a side effect the compiler hoists it, and
    the loop then spins on the registers of ONE read (round 6, the first build of skinny2_ks_kernel: found on the MI355X, invisible to the
    emulator).  The second number must be > 0 for every polling kernel."""
    out = {}
    for name, ins in kernels(lib).items():
        if name_filter and name_filter not in name: continue
        loads = [i for i, x in enumerate(ins) if x.startswith("buffer_load_dword") and " sc1" in x and " lds" not in x]
        if not loads: continue
        # backward branches: `s_cbranch_* N` with N >= 32768 (a 16-bit signed word offset printed unsigned) -> the loop spans [target, branch]
        inside = set()
        # instruction sizes are not in the text: approximate the loop body as "every instruction between the nearest preceding s_sleep and the
        # branch" -- the polling loops of this library all pause with s_sleep inside the loop
        for i, x in enumerate(ins):
            m = re.match(r"^s_cbranch_\w+ (\d+)", x)
            if not m or int(m.group(1)) < 32768: continue
            words = 65536 - int(m.group(1))
            j = i
            while j > 0 and words > 0:           # every instruction is 1 or 2 words: walking back `words` instructions over-covers, `words / 2` under-covers
                j -= 1; words -= 2
            inside.update(k for k in loads if j <= k <= i)
        out[name] = (len(loads), len(inside))
    return out


def packed_fp32(lib):
    """(number of packed fp32 arithmetic instructions, kernels that hold one) in the gfx950 code objects of `lib`.  The product build
    carries none (build.py FLAGS: -packed-fp32-ops off; profiles/r05_packed_fp32_hazard.md)."""
    n, names = 0, []
    for name, ins in kernels(lib).items():
        c = sum(1 for i in ins if re.match(r"v_pk_(mul|add|fma)_f32\b", i))
        if c:
            n += c; names.append(name)
    return n, names


def is_load(ins):
    return ins.startswith(("global_load", "buffer_load", "flat_load"))


def waits_inside_burst(ins, n_loads):
    """Indices of `s_waitcnt vmcnt(..)` that sit between the first and the n_loads-th global load of a kernel: a wait inside
    the request burst of a kernel that is meant to issue all of its requests back to back."""
    loads = [i for i, l in enumerate(ins) if is_load(l)]
    if len(loads) < n_loads: return None
    lo, hi = loads[0], loads[n_loads - 1]
    return [i for i in range(lo, hi) if re.match(r"s_waitcnt vmcnt\(\d+\)", ins[i])]


def scan(lib, window=40, max_n=1, name_filter=""):
    rows = []
    for name, ins in kernels(lib).items():
        if name_filter and name_filter not in name: continue
        loads = [i for i, l in enumerate(ins) if is_load(l)]
        hits = []
        for i, l in enumerate(ins):
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", l)
            if not m or int(m.group(1)) > max_n: continue
            before = sum(1 for j in loads if j < i)
            after = [j for j in loads if i < j <= i + window]
            if before and after: hits.append((i, int(m.group(1)), after[0] - i))
        if hits: rows.append((len(hits), name, len(ins), len(loads), hits[:6]))
    rows.sort(reverse=True)
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "qwen3-tts_amd", "libqtts.so"))
    ap.add_argument("--window", type=int, default=40); ap.add_argument("--max-n", type=int, default=1)
    ap.add_argument("--filter", default=""); ap.add_argument("--md", default=None)
    a = ap.parse_args()
    out = ["| serialized waits | kernel | instructions | loads | first hits (instr index, vmcnt, distance to next load) |", "|---|---|---|---|---|"]
    for n, name, ni, nl, hits in scan(a.lib, a.window, a.max_n, a.filter):
        out.append(f"| {n} | `{name[:110]}` | {ni} | {nl} | {hits} |")
    txt = "\n".join(out)
    print(txt)
    if a.md: open(a.md, "w").write(txt + "\n")
