#!/usr/bin/env bash
# Round 2, GPU call 16: waves per workgroup of skinny8_kernel (8 / 4 / 2 / 1), in-process A/B.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2j
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-12} "$OUT/$name.log" | cut -c1-400 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run ab 600 python tools/ab_inproc.py --frames 40 --reps 3
run pytest_subset 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "metric"
cat "$OUT/summary.txt"
