#!/usr/bin/env bash
# Round 3, GPU call 2: kernarg preload promoted to the default decode GEMM; attn_tk16 (talker decode attention on the matrix pipe,
# transposed V pages) vs attn_tk (QTTS_ATTN_MFMA=0) on the frame step and on 60 s utterances; full GPU suite; bench; in-kernel
# timestamps; rocprofv3 kernel trace of the bench command.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3b
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_gpu 1100 python -m pytest tests -q -m gpu -s
run frame_mfma_1 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --prof
QTTS_ATTN_MFMA=0 run frame_valu_1 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker
run frame_mfma_2 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker
QTTS_ATTN_MFMA=0 run frame_valu_2 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker
run long_mfma 400 python tools/bench_configs.py long --frames 750
QTTS_ATTN_MFMA=0 run long_valu 400 python tools/bench_configs.py long --frames 750
run bench 420 python bench.py --steps 5 --warmup 2
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
QTTS_LIBRARY=$PWD/qwen3-tts_amd/libqtts_tstamp.so TAILN=22 run ts_graph 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_graph.json"
( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o perf -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-roofline > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace.md" > /dev/null 2> "$OUT/rocpd_stats.err"; rm -rf "$OUT/prof"
cat "$OUT/rocpd_stats.err"; tail -14 "$OUT/kernel_trace.md"
grep -h "ms/frame" "$OUT"/frame_*.log
cat "$OUT/summary.txt"
