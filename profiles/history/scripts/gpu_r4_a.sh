#!/usr/bin/env bash
# Round 4, GPU call 1: baseline of the round-3 tree on this box (bench line with parity_mode) + the two codec kernel traces the round-3
# verdict found missing: B = 1 x 125 frames (BASELINE config 2) and B = 32 x 4 frames (the first-packet decode of config 4).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4a
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-900 | sed "s/^/    /"; }
trace() { local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_$name" -o perf -- python "$OLDPWD/tools/perf_frame.py" "$@" > "$OLDPWD/$OUT/rocprof_$name.log" 2>&1 ); echo "rocprof_$name rc=$?" | tee -a "$OUT/summary.txt"
  DB=$(find "$OUT/prof_$name" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/codec_kernel_trace_$name.md" > /dev/null 2>&1; rm -rf "$OUT/prof_$name"
  grep "^\[codec" "$OUT/rocprof_$name.log"; tail -2 "$OUT/codec_kernel_trace_$name.md"; }
: > "$OUT/summary.txt"
trace b1x125 --codec --reps 5 --batch 1 --codec-frames 125
trace b32x4 --codec --reps 5 --batch 32 --codec-frames 4
trace b8x125 --codec --reps 3 --batch 8 --codec-frames 125
run bench 420 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
run codec_only 200 python tools/bench_configs.py codec_only --trials 10
run first_packet 200 python tools/bench_configs.py first_packet --trials 10
cat "$OUT/summary.txt"
