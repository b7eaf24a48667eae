#!/usr/bin/env bash
# Round 4, GPU call 2: skinny8_f32_kernel (the parity mode's batch <= 8 decode GEMM) -- every fp32 real-dims golden, then the fp32 frame
# step new vs old (QTTS_SKINNY8F=0) and the 16-wave variant at K = 2048, then the bench line with parity_mode.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4b
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-900 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_f32 900 python -m pytest tests -q -m gpu -x -k "talker_06b or talker_17b or tiny_greedy or vs_oracle or prompt_assembly or large_batch"
run f32_new 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --talker-dtype f32 --reps 2
QTTS_SKINNY8F=0 run f32_old 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --talker-dtype f32 --reps 2
QTTS_SKINNY8F_NW=16 run f32_nw16 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --talker-dtype f32 --reps 2
QTTS_SKINNY8F_NW=4 run f32_nw4 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --talker-dtype f32 --reps 2
run bench 420 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
grep -h "greedy\|sampling" "$OUT"/f32_*.log | cut -c1-200
cat "$OUT/summary.txt"
