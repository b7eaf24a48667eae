#!/usr/bin/env bash
# Round 2, GPU call 2: the whole GPU suite on the new default build (skinny v2, promoted variants, teacher forcing), the frame
# timing, the phase ablation of the decode GEMM, a rocprofv3 kernel trace of the bench command, configs 2 and the long run.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2a
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n 4 "$OUT/$name.log" | cut -c1-300 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run pytest_gpu 900 python -m pytest tests -q -m gpu -s -x
run perf_frame 240 python tools/perf_frame.py --model 1.7b --frames 60 --talker --prof
run ablate 240 python tools/ablate_skinny.py
run bench 420 python bench.py --steps 3 --warmup 1
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o perf -- python "$OLDPWD/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof"
run codec_only 300 python tools/bench_configs.py codec_only --trials 10
run long 420 python tools/bench_configs.py long --frames 750
cat "$OUT/summary.txt"
