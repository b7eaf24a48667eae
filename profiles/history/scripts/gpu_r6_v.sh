#!/usr/bin/env bash
# Round 6, call v: split-K of the ring kernel (ticketed ordered combine): GPU tests, the A/B tool with the split tables.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6v
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
: > "$OUT/summary.txt"
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-400 | sed "s/^/    /"; }
run pytest_ring 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "ring_tap_gemm"
for i in 1 2; do run pytest_codec$i 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "codec or prefill"; done
TAILN=60 run bench_ring 900 python tools/bench_gemm_ring.py --screen 1
for B in 1 8; do timeout 300 python tools/perf_frame.py --codec --reps 5 --batch $B 2>&1 | grep "codec bf16" | tee -a "$OUT/summary.txt"; done
timeout 600 python tools/bench_configs.py first_packet --trials 10 2>&1 | tail -1 | cut -c1-700 | tee -a "$OUT/summary.txt"
