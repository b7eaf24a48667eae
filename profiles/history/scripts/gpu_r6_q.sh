#!/usr/bin/env bash
# Round 6, call q: the ring kernel, version 3 (branch-free steady step, MFMAs interleaved with the requests and reads)
# Linears), the codec at 1 x 10 s and 8 x 10 s with the ring on / off, the bf16 codec GPU tests.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6q
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
: > "$OUT/summary.txt"
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-400 | sed "s/^/    /"; }
run pytest_ring 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "ring_tap_gemm"
TAILN=40 run bench_ring 900 python tools/bench_gemm_ring.py --screen 2
for R in 0 1; do
  for B in 1 8; do
    QTTS_GEMM_RING=$R run codec_ring${R}_b$B 300 python tools/perf_frame.py --codec --reps 5 --batch $B
  done
done
run pytest_codec 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "codec"
cat "$OUT/summary.txt"
