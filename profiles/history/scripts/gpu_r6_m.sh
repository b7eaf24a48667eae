#!/usr/bin/env bash
# Round 6, call m: the bench line once more after the configs leg gave config 4 its own engine (bench.py only; kernels = the validated tree).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6m
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
timeout 900 python bench.py > "$OUT/bench_default.log" 2>&1; echo "bench_default rc=$?" | tee "$OUT/summary.txt"
grep -h '^{' "$OUT/bench_default.log" > "$OUT/bench_default.json"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "golden and 06b_one" > "$OUT/pytest_one.log" 2>&1; echo "pytest_one rc=$?" | tee -a "$OUT/summary.txt"
tail -c 600 "$OUT/bench_default.json"
