#!/usr/bin/env bash
# Round 6, call s: ring kernel v4 (slab-unrolled steady state for 7 / 2 taps): GPU tests, A/B table, ablation, codec timings.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6s
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
: > "$OUT/summary.txt"
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-400 | sed "s/^/    /"; }
run pytest_ring 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "ring_tap_gemm"
for i in 1 2; do run pytest_codec$i 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "codec"; done
TAILN=40 run bench_ring 900 python tools/bench_gemm_ring.py --screen 2
TAILN=40 run ablate 600 python tools/bench_gemm_ring.py --ablate
for R in 0 1; do for B in 1 8; do QTTS_GEMM_RING=$R run codec_ring${R}_b$B 300 python tools/perf_frame.py --codec --reps 5 --batch $B; done; done
cat "$OUT/summary.txt"
