#!/usr/bin/env bash
# Round 4, GPU call 9: 16 waves per workgroup for the bf16 decode GEMM's >= 16 MB operators (talker q|k|v, gate|up, down) -- the fp32
# kernel measured faster that way at K = 2048.  In-process interleaved A/B would be cleaner; the switch is read once, so: alternating processes.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4i
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-300 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
for r in 1 2 3; do
  run nw8_$r 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
  QTTS_SKINNY8_NW_BIG=16 run nw16_$r 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
done
QTTS_SKINNY8_NW_BIG=16 run nw16_prof 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 2 --prof
run nw8_prof 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 2 --prof
grep -h "sampling" "$OUT"/nw8_[123].log | cut -c1-120; echo ---; grep -h "sampling" "$OUT"/nw16_[123].log | cut -c1-120
echo "--- nw8 per class"; grep -h "stack 0" "$OUT"/nw8_prof.log; echo "--- nw16 per class"; grep -h "stack 0" "$OUT"/nw16_prof.log
