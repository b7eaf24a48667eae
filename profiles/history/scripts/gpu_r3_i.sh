#!/usr/bin/env bash
# Round 3, GPU call 9: single-wave sampler for the code predictor (sample_kernel_w1) vs sample_kernel_v2 (QTTS_SAMPLER_W1=0) on the frame
# step; full GPU suite on the build; bench; kernel trace.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3i
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_gpu 1100 python -m pytest tests -q -m gpu -s
run frame_w1_1 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker
QTTS_SAMPLER_W1=0 run frame_v2_1 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker
run frame_w1_2 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker
QTTS_SAMPLER_W1=0 run frame_v2_2 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker
grep -h "ms/frame" "$OUT"/frame_*.log | cut -c1-90
run bench 420 python bench.py --steps 5 --warmup 2
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o perf -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-roofline > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof"
grep -E "sample_kernel|attn_cp|embed_sum" "$OUT/kernel_trace.md"
cat "$OUT/summary.txt"
