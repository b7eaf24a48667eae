#!/usr/bin/env bash
# Round 4, GPU call 24: the fused launch with its first operands as preloaded kernel arguments -- frame step, timeline; then the whole GPU
# suite, smoke() and the bench line on this tree (attention.hip changed; the stamped decode-GEMM passes of call 20 still describe skinny.hip /
# talker_engine.hip bit for bit).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4w
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-700 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run frame_front_1 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
QTTS_CP_ATTN_O=0 run frame_plain_1 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
run frame_front_2 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
TAILN=30 QTTS_LIBRARY_OK=1 run ts_front 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_front.json"
TAILN=4 run pytest_gpu 900 python -m pytest tests -q -m gpu
run smoke 200 python __graft_entry__.py --smoke
run bench 900 python bench.py --steps 20 --warmup 5
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
for f in "$OUT"/frame_*.log; do echo "$(basename $f): $(grep -h sampling $f | cut -c1-120)"; done
grep -h "cp_attn_o" "$OUT"/ts_front.log | cut -c1-250
cat "$OUT/summary.txt"
