#!/usr/bin/env bash
# Round 2, GPU call 6: codec GEMM epilogue v2 + tap-reuse kernel variants, codec tests, MFMA-busy counters.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2d
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n 3 "$OUT/$name.log" | cut -c1-400 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run pytest_codec 600 python -m pytest tests -q -m gpu -s -k "codec or tokenizer or wrapper or encoder or speaker"
run codec_tap2 300 python tools/bench_configs.py codec_only --trials 10
run codec_tap2_bk32 300 env QTTS_TAP2_BK=32 python tools/bench_configs.py codec_only --trials 10
run codec_old 300 env QTTS_CODEC_FAST16=0 python tools/bench_configs.py codec_only --trials 10
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o perf -- python "$OLDPWD/tools/perf_frame.py" --codec --reps 1 --batch 8 > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/codec_kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof"
cat "$OUT/summary.txt"
