#!/usr/bin/env bash
# Round 6, call y: race screen of the ring kernel under memory contention (bitwise against gemm_dma), then the codec / prefill GPU tests five times.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6y
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
: > "$OUT/summary.txt"
timeout 900 python tools/ring_race_screen.py --runs 150 > "$OUT/race_screen.log" 2>&1; echo "race_screen rc=$?" | tee -a "$OUT/summary.txt"; tail -10 "$OUT/race_screen.log"
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "codec or prefill or ring" > "$OUT/pytest_$i.log" 2>&1; echo "pytest_$i rc=$? $(tail -1 $OUT/pytest_$i.log)" | tee -a "$OUT/summary.txt"; done
