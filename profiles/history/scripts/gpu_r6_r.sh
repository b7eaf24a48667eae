#!/usr/bin/env bash
# Round 6, call r: ring kernel v3 after the drain-wait fix: the ring / codec GPU tests, then the ablation table of the steady step.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6r
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
: > "$OUT/summary.txt"
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-400 | sed "s/^/    /"; }
run pytest_ring 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "ring_tap_gemm"
for i in 1 2 3; do run pytest_codec$i 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "codec"; done
TAILN=40 run ablate 600 python tools/bench_gemm_ring.py --ablate
cat "$OUT/summary.txt"
