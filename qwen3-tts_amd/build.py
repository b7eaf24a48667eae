"""Build libqtts.so (the C-ABI HIP library) in-tree for gfx950.

    python qwen3-tts_amd/build.py [--force] [--variant NAME | --all-variants]

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the tree.

Variants (A/B material, never the default): the same sources with extra -D flags, built into `libqtts_<name>.so`
next to the product library and selected at run time with `QTTS_LIBRARY=<path>` (see tools/ab_variants.py).  The
product library is always the flag-free build; a variant becomes the default only by moving its code under the
default branch of the source after it has been measured on hardware.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libqtts.so")
SOURCES = ["gemm_tap.hip", "resunit.hip", "skinny.hip", "elementwise.hip", "attention.hip", "cp_mlp.hip", "cp_mlp32.hip", "cp_layer.hip", "sampling.hip",
           "codec_engine.hip", "talker_engine.hip", "encoder_kernels.hip", "encoder_engine.hip",
           "speaker_kernels.hip", "speaker_engine.hip", "stream_kernels.hip"]
# sources that only a measuring variant links (never the product library): variant name -> files
VARIANT_SOURCES = {"probe": ["persist_probe.hip"]}
HEADERS = ["common.h", "kernels.h", "glue.h", "granule.h", "attn_helpers.h", "tstamp.h", os.path.join("..", "..", "include", "qtts.h")]
# -amdgpu-kernarg-preload-count: the leading scalar kernel arguments (14 dwords on gfx950) arrive in user SGPRs with the wave instead
# of behind an `s_load` round trip; the frame step's decode GEMMs (skinny8_kernel, skinny8_f32_kernel) pass their address operands that way.
# The flag applies to every kernel of the library (only leading SCALAR arguments are ever preloaded; a kernel whose first argument is a
# by-value struct is unaffected) and needs a gfx940+ firmware / ROCm >= 6.1 that implements kernarg preload -- true of every MI355X stack.
# -target-feature -packed-fp32-ops (round 5): NO packed fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) in the device code.
# With them, a sequence the compiler forms out of scalar source -- attn_cp's RoPE: `v_pk_mul_f32 v[8:9], v[8:9], v[6:7] op_sel:[0,1]
# op_sel_hi:[0,0]` ... `v_sub_f32 v2, v14, v8` -- returned, in lanes 48-63 of the wave and only while another stream's kernels shared the
# compute units, x0' c instead of x0' c - x1' sn: identical inputs, operands long arrived (`s_waitcnt vmcnt(0)` thirty instructions before),
# 13 of 32 generations affected; 0 of 32 with this flag (profiles/r05_packed_fp32_hazard.md: the bisection from "codes differ under a
# concurrent codec decode" down to the instruction pair).  The library had 16 275 such instructions in 318 kernels; none of the hot loops
# is VALU-bound (the frame step is launch-latency-bound, the codec MFMA- and LDS-bound): the measured cost is on the same page.
# (The host pass of hipcc reports the feature as unknown for x86 and ignores it.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-mllvm", "-amdgpu-kernarg-preload-count=16",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


# name -> extra compiler flags; what each one tests is written next to the macro in the source.
VARIANTS = {
    # attention.hip: KV beyond the 256-key prefetch window read 4 chunks per latency round instead of 1.  Only long
    # sequences see it: A/B with `tools/ab_variants.py --frames 600 --only default attn_tail` (S grows to ~650).
    "attn_tail": ["-DQTTS_ATTN_TAIL_BATCH=1"],
    # csrc/tstamp.h: phase timestamps inside the frame step's kernels (decode GEMM, both decode attentions, sampler); a
    # measuring build for tools/ts_frame.py, never the product.
    "tstamp": ["-DQTTS_TSTAMP=1"],
    # skinny.hip: the perf-ablation branches of the decode GEMM (`SkinnyParams::ablate`: no done check / no x loads / no epilogue
    # loads / no weight stream) exist only in this build (tools/ablate_skinny.py); the product build has them compiled out.
    "ablate": ["-DQTTS_ABLATE=1"],
    # csrc/persist_probe.hip (hand-off probe: persistent launch with grid barriers vs graph launches, tools/persist_probe.py):
    # the product sources + the probe's own file and entry point; round 3 linked it into libqtts.so, round 4 moved it here.
    "probe": ["-DQTTS_PROBE=1"],
    # Round 5 diagnosis (profiles/r05_packed_fp32_hazard.md): the same sources WITH packed fp32 instructions (the compiler's default for gfx950;
    # the product build disables them, FLAGS above) -- the build that reproduces the run-to-run differences under a concurrent codec decode.
    "pk": ["-Xclang", "-target-feature", "-Xclang", "+packed-fp32-ops"],
    # Round 6, gemm_ring_kernel's unrolled step: "1 MFMA, then up to N others" -- N = 3 in the product; 0 = no sched_group_barrier hints at all
    # (tools/bench_gemm_ring.py with QTTS_LIBRARY; profiles/r06_gemm_ring.md)
    "ring_sgb0": ["-DQTTS_RING_SGB_ALT=1"], "ring_sgb2": ["-DQTTS_RING_SGB_ALT=2"], "ring_sgb4": ["-DQTTS_RING_SGB_ALT=4"],
}
# Round 3: kpre (kernarg preload for the decode GEMM) measured 0.973x per frame (profiles/r03_ab_kpre.md) and is now the default code.
# Round 2 (profiles/r02_ab_variants.md): cp_pretable, cp_qkvtable, attn_cp and sampler_v2 were measured faster and are now the
# default code; attn_t1, wtemporal, late_norm, embed_sum_v2 and gu8 were measured slower or neutral and are deleted.


def _digest(paths, extra=()):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + list(extra)).encode())
    return h.hexdigest()


def variant_path(variant: str) -> str:
    return os.path.join(HERE, f"libqtts_{variant}.so")


# The toolchain the flags above and the ISA-level assumptions of the sources were validated on: the packed-fp32 hazard's flag
# (tests/test_gpu_parity.py: the contention test; tests/test_host_logic.py: no v_pk_*_f32 in the code objects), the polling loads that must stay
# inside their loops, the barrier / LDS-DMA wait placement and gemm_ring_kernel's counted waits and MFMA interleave (all pinned from the code
# objects in tests/test_host_logic.py).  Another compiler may schedule differently: the build says so, the ISA tests decide.
VALIDATED_TOOLCHAIN = "roc-7.2.0"


def toolchain_banner(hipcc: str) -> str:
    try:
        return subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception as e:      # noqa: BLE001 -- a missing compiler fails at the first compile with its own message
        return f"(hipcc --version failed: {e})"


def build(force: bool = False, verbose: bool = True, variant: str = None) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if VALIDATED_TOOLCHAIN not in toolchain_banner(hipcc):
        print(f"[build] WARNING: {hipcc} is not the toolchain these sources were validated on ({VALIDATED_TOOLCHAIN}): re-run the ISA pins "
              f"(pytest tests/test_host_logic.py -k 'isa or waits or packed or polling or dma') and the GPU contention test before trusting the library", file=sys.stderr)
    extra = []
    out = OUT
    if variant is not None:
        if variant not in VARIANTS:
            raise ValueError(f"unknown variant {variant!r}; known: {sorted(VARIANTS)}")
        extra = VARIANTS[variant]
        out = variant_path(variant)
    objdir = os.path.join(HERE, "build" if variant is None else f"build_{variant}")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for s in SOURCES + (VARIANT_SOURCES.get(variant, []) if variant else []):
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        stamp = obj + ".sha"
        dig = _digest([src] + hdrs, extra)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), stamp, dig, s))
    failed = []
    for p, stamp, dig, s in procs:
        if p.wait() != 0:
            failed.append(s)
        else:
            with open(stamp, "w") as f:
                f.write(dig)
    if failed:
        raise RuntimeError(f"hipcc failed for: {failed}")
    if procs or not os.path.exists(out) or force:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--all-variants" in sys.argv:
        for v in sorted(VARIANTS):
            print(build(force=force, variant=v))
    elif "--variant" in sys.argv:
        print(build(force=force, variant=sys.argv[sys.argv.index("--variant") + 1]))
    else:
        print(build(force=force))
