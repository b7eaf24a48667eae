#!/usr/bin/env bash
# Round 4, GPU call 18: the code predictor's q|k|v GEMM in front of attention + o-projection in the same launch (layers >= 1) -- frame step
# A/B (front on / attention + o-projection only / separate launches), the reducer's first-read delay, timeline, GPU tests.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4q
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run frame_front_0 100 python tools/perf_frame.py --model 1.7b --frames 30 --talker --reps 1
TAILN=6 run pytest_fused 600 python -m pytest tests -q -m gpu -x -s -k "fused_attention or bf16_mode_pinned or tiny_greedy"
for i in 1 2; do
  run frame_front_$i 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
  QTTS_CP_FRONT=0 run frame_attno_$i 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
  QTTS_CP_ATTN_O=0 run frame_plain_$i 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
done
QTTS_CP_ATTN_O_PAUSE=12 run frame_pause12 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
QTTS_CP_ATTN_O_PAUSE=28 run frame_pause28 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
TAILN=30 QTTS_LIBRARY_OK=1 run ts_front 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_front.json"
for f in "$OUT"/frame_*.log; do echo "$(basename $f): $(grep -h sampling $f | cut -c1-120)"; done
grep -h "cp_attn_o" "$OUT"/pytest_fused.log "$OUT"/ts_front.log | cut -c1-250
cat "$OUT/summary.txt"
