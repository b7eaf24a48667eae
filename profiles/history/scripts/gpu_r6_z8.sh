#!/usr/bin/env bash
# Round 6, call z8: the GPU suite after the checker-side thread fix (tests/test_gpu_parity.py `_one_thread`: the tiny-tensor oracle work of the
# sampled-path and the long-generation tests on one torch thread; 330 + 71 s of the suite's 765-800 s were thread-pool hand-offs), with per-test
# durations; smoke; the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6z8
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=4 run pytest_gpu 1700 python -m pytest tests -q -m gpu -s --durations=25
run smoke 200 python __graft_entry__.py --smoke
run bench 900 python bench.py --steps 20 --warmup 5
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
grep -E "^[0-9.]+s (call|setup)" "$OUT/pytest_gpu.log" | head -12
cat "$OUT/summary.txt"
