#!/usr/bin/env bash
# Round 2, GPU call 21: gemm_tap2 with weight tiles requested two steps ahead -- codec timing + parity.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2m
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-4} "$OUT/$name.log" | cut -c1-250 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run perf_codec 300 python tools/perf_frame.py --model 1.7b --codec --codec-frames 125 --reps 7
run perf_codec_b1 300 python tools/perf_frame.py --model 1.7b --codec --codec-frames 125 --reps 7 --batch 1
run pytest_codec 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "codec"
cat "$OUT/summary.txt"
