#!/usr/bin/env bash
# Round 4, GPU call 7: the codec's <= 128-row transformer as weight-streaming strips (skinny2 with eight m-tiles) and decoder.0 through the
# bf16-activation tap GEMM -- codec tests, config 2 / 8 x 10 s / first packet with each switched off, kernel trace at B = 1 x 125.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4g
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-1200 | sed "s/^/    /"; }
trace() { local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_$name" -o perf -- python "$OLDPWD/tools/perf_frame.py" "$@" > "$OLDPWD/$OUT/rocprof_$name.log" 2>&1 ); echo "rocprof_$name rc=$?" | tee -a "$OUT/summary.txt"
  DB=$(find "$OUT/prof_$name" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/codec_kernel_trace_$name.md" > /dev/null 2>&1; rm -rf "$OUT/prof_$name"
  grep "^\[codec" "$OUT/rocprof_$name.log"; head -14 "$OUT/codec_kernel_trace_$name.md" | cut -c1-120; tail -1 "$OUT/codec_kernel_trace_$name.md"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_codec 600 python -m pytest tests -q -m gpu -x -s -k "codec or wrapper or stream or smoke or from_pretrained"
run codec_dflt 200 python tools/bench_configs.py codec_only --trials 10
QTTS_CODEC_SKINNY=0 run codec_noskinny 200 python tools/bench_configs.py codec_only --trials 10
QTTS_CODEC_DEC0_A16=0 run codec_nodec0 200 python tools/bench_configs.py codec_only --trials 10
run fp_dflt 200 python tools/bench_configs.py first_packet --trials 10
QTTS_CODEC_SKINNY=0 QTTS_CODEC_DEC0_A16=0 run fp_off 200 python tools/bench_configs.py first_packet --trials 10
trace b1x125 --codec --reps 5 --batch 1 --codec-frames 125
trace b32x4 --codec --reps 5 --batch 32 --codec-frames 4
cat "$OUT/summary.txt"
