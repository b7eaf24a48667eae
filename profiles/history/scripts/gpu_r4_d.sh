#!/usr/bin/env bash
# Round 4, GPU call 4: fp32 split-K (producer halves + combining consumer) -- every fp32 golden, frame time on / off, per class;
# codec: decode calls as hipGraphs and 64-wide k-slabs for under-filled grids -- codec tests, config 2 and the first packet, each A/B.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4d
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-1200 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_f32_codec 900 python -m pytest tests -q -m gpu -x -k "talker_06b or talker_17b or tiny_greedy or vs_oracle or prompt_assembly or large_batch or codec or wrapper or stream"
run f32_split 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --talker-dtype f32 --reps 2 --prof
QTTS_SKINNY8F_SPLITK=0 run f32_nosplit 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --talker-dtype f32 --reps 2
run codec_dflt 200 python tools/bench_configs.py codec_only --trials 10
QTTS_CODEC_GRAPH=0 run codec_nograph 200 python tools/bench_configs.py codec_only --trials 10
QTTS_TAP2_BK=32 run codec_bk32 200 python tools/bench_configs.py codec_only --trials 10
run fp_dflt 200 python tools/bench_configs.py first_packet --trials 10
QTTS_CODEC_GRAPH=0 QTTS_TAP2_BK=32 run fp_r3 200 python tools/bench_configs.py first_packet --trials 10
run bench 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api-e2e
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
grep -h "greedy\|sampling" "$OUT"/f32_*.log | cut -c1-200
grep -h "stack\|decode GEMM" "$OUT"/f32_split.log | cut -c1-200
cat "$OUT/summary.txt"
