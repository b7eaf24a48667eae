#!/usr/bin/env bash
# Round 2, GPU call 3: masked grouped loads + straight-line chunk variants + next-weights warm-up in the decode GEMM.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2b
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n 4 "$OUT/$name.log" | cut -c1-300 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run perf_frame_pf1 240 python tools/perf_frame.py --model 1.7b --frames 60 --talker --prof
run perf_frame_pf0 240 env QTTS_SKINNY_PREFETCH=0 python tools/perf_frame.py --model 1.7b --frames 60 --talker --prof
run perf_frame_pf1_again 240 python tools/perf_frame.py --model 1.7b --frames 60 --talker
run ablate 240 python tools/ablate_skinny.py
run pytest_gpu 900 python -m pytest tests -q -m gpu -s
cat "$OUT/summary.txt"
