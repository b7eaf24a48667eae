#!/usr/bin/env python3
"""A/B harness for the frame step (GPU box): the product library against build variants and HIP-runtime knobs.

Each candidate runs `tools/perf_frame.py --talker` (1.7B dims, batch 8, hipGraph) in its OWN process under a timeout,
with either `QTTS_LIBRARY=<variant .so>` (qwen3-tts_amd/build.py VARIANTS, built here on the CPU container so that
they travel with the snapshot) or extra environment variables for the runtime.  Prints one table and writes
gpurun_out/ab/ab.json.  Nothing here changes the product: a variant that wins is promoted by hand after the numbers
are in profiles/.

    python qwen3-tts_amd/build.py --all-variants          # CPU container, before gpurun
    gpurun --timeout 900 -- 'python tools/ab_variants.py --frames 40'
    gpurun --timeout 1500 -- 'python tools/ab_variants.py --check combo --only default combo default_again'
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-tts_amd"))
import build as qbuild  # noqa: E402

# runtime knobs present in this image's libamdhip64.so (strings); values that differ from what we believe is the default
ENV_CANDIDATES = {
    "graph_packet_capture_off": {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"},
    "graph_packet_capture_on": {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"},
    "dev_kernarg_off": {"HIP_FORCE_DEV_KERNARG": "0"},
    "dev_kernarg_on": {"HIP_FORCE_DEV_KERNARG": "1"},
    "graph_batch_1": {"DEBUG_HIP_GRAPH_BATCH_SIZE": "1"},
}


def run_one(name, env_extra, frames, model, timeout):
    env = dict(os.environ)
    env.update(env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "perf_frame.py"), "--model", model, "--frames", str(frames),
           "--talker", "--reps", "3"]
    t0 = time.time()
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        text = out.stdout + out.stderr
        rc = out.returncode
    except subprocess.TimeoutExpired as e:
        text = (e.stdout or "") + (e.stderr or "") if isinstance(e.stdout, str) else "timeout"
        rc = -9
    res = {"name": name, "env": env_extra, "rc": rc, "wall_s": round(time.time() - t0, 1)}
    for mode in ("sampling", "greedy"):
        m = re.search(r"\[%s\].*?([0-9.]+) ms/frame" % mode, text)
        if m:
            res[mode + "_ms_per_frame"] = float(m.group(1))
    if rc != 0:
        res["tail"] = text[-600:]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--model", default="1.7b")
    ap.add_argument("--timeout", type=int, default=240, help="per candidate, seconds")
    ap.add_argument("--only", nargs="*", default=None, help="candidate names to run (default: all)")
    ap.add_argument("--check", nargs="*", default=None, metavar="VARIANT",
                    help="before timing, run the talker GPU parity tests (reference goldens, bit-exact greedy codes) against "
                         "these library variants (no names = every built variant); a variant that fails is not timed")
    args = ap.parse_args()
    cands = [("default", {})]
    for v in sorted(qbuild.VARIANTS):
        path = qbuild.variant_path(v)
        if os.path.exists(path):
            cands.append(("lib:" + v, {"QTTS_LIBRARY": path}))
        else:
            print(f"[ab] variant {v}: {path} not built (python qwen3-tts_amd/build.py --all-variants)", file=sys.stderr)
    cands += [("env:" + k, v) for k, v in ENV_CANDIDATES.items()]
    cands.append(("default_again", {}))                 # drift check: first and last run are the same thing
    if args.only:
        cands = [c for c in cands if c[0] in args.only or c[0].split(":", 1)[-1] in args.only]
    results = []
    if args.check is not None:
        failed = set()
        for name, env_extra in list(cands):
            if not name.startswith("lib:") or (args.check and name[4:] not in args.check):
                continue
            env = dict(os.environ, **env_extra)
            t0 = time.time()
            try:
                out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x",
                                      "-m", "gpu", "-k", "talker or sampler or prompt_assembly"], env=env, capture_output=True,
                                     text=True, timeout=600, cwd=ROOT)
                rc, tail = out.returncode, (out.stdout + out.stderr)[-400:]
            except subprocess.TimeoutExpired:
                rc, tail = -9, "timeout"
            print(f"[ab] parity {name:28s} rc={rc} ({time.time() - t0:.0f} s)", flush=True)
            results.append({"name": "parity:" + name, "rc": rc, "tail": tail if rc else ""})
            if rc != 0:
                failed.add(name)
        cands = [c for c in cands if c[0] not in failed]
    for name, env in cands:
        r = run_one(name, env, args.frames, args.model, args.timeout)
        results.append(r)
        print(f"[ab] {name:32s} rc={r['rc']:3d} sampling={r.get('sampling_ms_per_frame')} "
              f"greedy={r.get('greedy_ms_per_frame')} ({r['wall_s']} s)", flush=True)
    base = next((r.get("sampling_ms_per_frame") for r in results if r["name"] == "default"), None)
    print("\n| candidate | sampling ms/frame | vs default | greedy ms/frame |\n|---|---|---|---|")
    for r in results:
        if r["name"].startswith("parity:"):
            continue
        s = r.get("sampling_ms_per_frame")
        rel = f"{s / base:.3f}x" if (s and base) else "-"
        print(f"| {r['name']} | {s} | {rel} | {r.get('greedy_ms_per_frame')} |")
    out_dir = os.path.join(ROOT, "gpurun_out", "ab")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "ab.json"), "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
