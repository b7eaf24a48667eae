#!/usr/bin/env bash
# Round 3, GPU call 17 (call 16 died on an uninitialised attention-parameter field): validation of the small-grid GEMM work (tile chooser, 256-wide k-steps, bf16 hand-over in the prefill, bf16
# hand-over to the final convolution): full GPU suite, bench line, frame / prefill A/B, configs 2 and 4, GEMM microbenchmark.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3q
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_gpu 1100 python -m pytest tests -q -m gpu -s
run bench 420 python bench.py --gpus 1 --steps 10 --warmup 3
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
for i in 1; do
  run frame_a16_$i 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker
  QTTS_PREFILL_A16=0 run frame_a32_$i 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker
  QTTS_PREFILL_A16=0 QTTS_GEMM_NARROW=0 run frame_r2_$i 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker
done
grep -h "prefill+1tok" "$OUT"/frame_*.log | cut -c1-90
run codec_only 120 python tools/bench_configs.py codec_only
run first_packet 200 python tools/bench_configs.py first_packet
QTTS_PREFILL_A16=0 QTTS_GEMM_NARROW=0 QTTS_GEMM_WIDE_MAX=1024 run first_packet_r2 200 python tools/bench_configs.py first_packet
run gemm_small 300 python tools/bench_gemm_small.py
