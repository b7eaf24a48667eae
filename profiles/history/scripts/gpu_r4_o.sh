#!/usr/bin/env bash
# Round 4, GPU call 16: cp_attn_o third version (256 workgroups of 4 waves: one attention unit per SIMD; tagged granules) -- GPU tests (run-to-run identical, bf16 pin),
# the in-kernel timeline, the frame step A/B in alternating processes.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4o
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
QTTS_CP_ATTN_O=1 run frame_fused_0 100 python tools/perf_frame.py --model 1.7b --frames 30 --talker --reps 1
TAILN=6 run pytest_fused 600 python -m pytest tests -q -m gpu -x -s -k "fused_attention or bf16_mode_pinned or tiny_greedy"
for i in 1 2; do
  QTTS_CP_ATTN_O=1 run frame_fused_$i 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
  QTTS_CP_ATTN_O=0 run frame_plain_$i 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
done
TAILN=30 QTTS_LIBRARY_OK=1 QTTS_CP_ATTN_O=1 run ts_fused 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_fused.json"
grep -h "sampling\|greedy" "$OUT"/frame_*.log | cut -c1-170
grep -h "cp_attn_o" "$OUT"/pytest_fused.log "$OUT"/ts_fused.log | cut -c1-250
cat "$OUT/summary.txt"
