#!/usr/bin/env python3
"""Static resource table of every gfx950 kernel in libqtts.so (no GPU needed).

Reads the AMDGPU metadata note of each embedded code object (llvm-objdump --offloading + llvm-readelf --notes)
and prints, per kernel: VGPR / AGPR / SGPR counts, spills, scratch bytes, static LDS bytes, the workgroup size
bound, and the occupancy those numbers allow on a CDNA4 CU (512 unified VGPR+AGPR registers per SIMD lane,
allocation granule 8, at most 8 waves per SIMD; 160 KB LDS per CU).  Kernels launched with dynamic LDS get the
host-side request added by hand in DYNAMIC_LDS below (the launchers raise hipFuncAttributeMaxDynamicSharedMemorySize
themselves when the request exceeds 64 KB).

What it is for: the things the host SIMT emulator (tests/hostemu) cannot see -- register spills, scratch use, LDS
capacity, occupancy -- checked before a kernel's first hardware run.  `--check` exits non-zero if any kernel
spills, uses scratch, or exceeds the LDS of a CU.

    python tools/kernel_resources.py [--so qwen3-tts_amd/libqtts.so] [--md profiles/rXX_kernel_resources.md] [--check]
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LDS_PER_CU = 160 * 1024
VGPR_FILE = 512          # unified VGPR + AGPR registers per lane per SIMD (gfx90a and later)
VGPR_GRANULE = 8
MAX_WAVES_PER_SIMD = 8


def demangle(names):
    tool = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")
    try:
        out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
    except OSError:
        return names
    return out.stdout.splitlines() if out.returncode == 0 else names


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)          # drop the parameter list
    return name.replace("qtts::", "")


def kernels_of(so):
    tmp = tempfile.mkdtemp(prefix="qtts_co_")
    try:
        local = os.path.join(tmp, os.path.basename(so))
        shutil.copy(so, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True,
                       capture_output=True)
        rows = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)],
                                   capture_output=True, text=True, check=True).stdout
            m = re.search(r"^\s*---\n(.*?)^\s*\.\.\.", notes, re.S | re.M)
            if not m:
                continue
            meta = yaml.safe_load(m.group(1))
            for k in meta.get("amdhsa.kernels", []):
                rows.append(k)
        return rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def occupancy(k):
    regs = k[".vgpr_count"]          # (gfx90a+ unified file: the total -- arch registers + the accumulator registers `.agpr_count` reports separately)
    regs = max(VGPR_GRANULE, -(-regs // VGPR_GRANULE) * VGPR_GRANULE)
    by_regs = min(MAX_WAVES_PER_SIMD, VGPR_FILE // regs)
    wg_waves = -(-k[".max_flat_workgroup_size"] // 64)
    lds = k[".group_segment_fixed_size"]
    by_lds = MAX_WAVES_PER_SIMD if lds == 0 else (LDS_PER_CU // lds) * wg_waves / 4.0
    return by_regs, min(by_regs, by_lds)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "qwen3-tts_amd", "libqtts.so"))
    ap.add_argument("--md", default=None, help="also write the table as markdown to this file")
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    ks = kernels_of(args.so)
    names = [short(n) for n in demangle([k[".name"] for k in ks])]
    lines = ["| kernel | wg | vgpr | agpr | sgpr | spills v/s | scratch B | static LDS B | waves/SIMD (regs) |",
             "|---|---|---|---|---|---|---|---|---|"]
    bad = []
    for n, k in sorted(zip(names, ks), key=lambda t: t[0]):
        by_regs, _ = occupancy(k)
        spills = (k.get(".vgpr_spill_count", 0), k.get(".sgpr_spill_count", 0))
        scratch = k.get(".private_segment_fixed_size", 0)
        lds = k[".group_segment_fixed_size"]
        lines.append(f"| `{n}` | {k['.max_flat_workgroup_size']} | {k['.vgpr_count']} | {k.get('.agpr_count', 0)} | "
                     f"{k['.sgpr_count']} | {spills[0]}/{spills[1]} | {scratch} | {lds} | {by_regs} |")
        if spills[0] or scratch or lds > LDS_PER_CU or k.get(".uses_dynamic_stack"):
            bad.append(n)
    text = "\n".join(lines)
    print(text)
    print(f"\n{len(ks)} kernels; {len(bad)} with spills / scratch / oversize LDS" + (": " + ", ".join(bad) if bad else ""))
    if args.md:
        with open(args.md, "w") as f:
            f.write("# Static kernel resources (gfx950 code objects of libqtts.so; `python tools/kernel_resources.py`)\n\n"
                    "From the AMDGPU metadata notes -- no hardware involved.  waves/SIMD = min(8, 512 / (vgpr + agpr "
                    "rounded up to 8)).\n\n" + text + f"\n\n{len(ks)} kernels; {len(bad)} with spills / scratch / "
                    "oversize static LDS" + (": " + ", ".join(bad) if bad else "") + ".\n")
    if args.check and bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
