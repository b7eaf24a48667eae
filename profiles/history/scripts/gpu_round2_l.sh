#!/usr/bin/env bash
# Round 2, GPU call 18: validation of the round's final default build -- full GPU suite, bench line, rocprofv3 kernel trace of the
# bench command, FETCH_SIZE pass of the decode GEMM, in-kernel timestamps, long utterance, codec config 2, 0.6B frame.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2l
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-400 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run pytest_gpu 900 python -m pytest tests -q -m gpu -s
run bench 420 python bench.py --steps 5 --warmup 2
run perf_frame 240 python tools/perf_frame.py --model 1.7b --frames 60 --talker --prof
run perf_frame_06b 240 python tools/perf_frame.py --model 0.6b --frames 60 --talker
run long 420 python tools/bench_configs.py long --frames 750
run codec_only 300 python tools/bench_configs.py codec_only --trials 10
TAILN=22 run ts_graph 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_graph.json"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o perf -- python "$OLDPWD/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/$OUT/pmc1" -o pmc -- python "$OLDPWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph > "$OLDPWD/$OUT/pmc_fetch.log" 2>&1 ); echo "pmc_fetch rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/pmc1" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_fetch_size.md" > /dev/null 2>&1; rm -rf "$OUT/pmc1"
cat "$OUT/summary.txt"
