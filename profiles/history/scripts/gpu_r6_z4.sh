#!/usr/bin/env bash
# Round 6, call z4: the tree with the scheduling hint at 3: ring / codec / prefill GPU tests, race screen, codec timings, the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6z4
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
: > "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "codec or prefill or ring" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)" | tee -a "$OUT/summary.txt"
timeout 900 python tools/ring_race_screen.py --runs 100 > "$OUT/race_screen.log" 2>&1; echo "race_screen rc=$? $(tail -1 $OUT/race_screen.log)" | tee -a "$OUT/summary.txt"
for B in 1 8; do timeout 300 python tools/perf_frame.py --codec --reps 5 --batch $B 2>&1 | grep "codec bf16" | tee -a "$OUT/summary.txt"; done
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.log" 2>&1; echo "bench rc=$?" | tee -a "$OUT/summary.txt"
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
