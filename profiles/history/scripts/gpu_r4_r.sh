#!/usr/bin/env bash
# Round 4, GPU call 19: two reads in flight at both hand-offs of cp_attn_o -- frame step over (first pause, step), against attention +
# o-projection only and the separate launches; timeline; GPU tests.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4r
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run frame_p16s8_0 100 python tools/perf_frame.py --model 1.7b --frames 30 --talker --reps 1
TAILN=6 run pytest_fused 600 python -m pytest tests -q -m gpu -x -s -k "fused_attention or tiny_greedy"
run frame_p16s8_1 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
QTTS_CP_ATTN_O=0 run frame_plain_1 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
QTTS_CP_FRONT=0 run frame_attno_1 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
QTTS_CP_ATTN_O_PAUSE=8 run frame_p8s8 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
QTTS_CP_ATTN_O_PAUSE=24 run frame_p24s8 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
QTTS_CP_ATTN_O_PAUSE=16 QTTS_CP_ATTN_O_STEP=4 run frame_p16s4 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
QTTS_CP_ATTN_O_PAUSE=32 QTTS_CP_ATTN_O_STEP=8 run frame_p32s8 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
run frame_p16s8_2 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
TAILN=30 QTTS_LIBRARY_OK=1 run ts_front 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_front.json"
for f in "$OUT"/frame_*.log; do echo "$(basename $f): $(grep -h sampling $f | cut -c1-120)"; done
grep -h "cp_attn_o" "$OUT"/pytest_fused.log "$OUT"/ts_front.log | cut -c1-250
cat "$OUT/summary.txt"
