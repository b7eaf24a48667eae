#!/usr/bin/env bash
# Round 6, call z5: same-box A/B of the whole codec decode with the scheduling hint at 3 (product) and at 2 (build variant ring_sgb2), interleaved.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6z5
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
: > "$OUT/summary.txt"
for rep in 1 2 3; do for v in "" _ring_sgb2; do for B in 1 8; do
  echo -n "libqtts$v.so rep $rep: " | tee -a "$OUT/summary.txt"; QTTS_LIBRARY="$PWD/qwen3-tts_amd/libqtts$v.so" timeout 300 python tools/perf_frame.py --codec --reps 7 --batch $B 2>&1 | grep "codec bf16" | tee -a "$OUT/summary.txt"
done; done; done
