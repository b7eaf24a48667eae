#!/usr/bin/env bash
# Round 6, call z3: refresh of the side measurements on the final tree: pure frame step (batch 8 and 32), the 60 s utterance, config 2 as its own job.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6z3
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
: > "$OUT/summary.txt"
timeout 600 python tools/perf_frame.py --model 1.7b --frames 125 --talker --reps 3 2>&1 | grep "^\[" | tee -a "$OUT/summary.txt"
timeout 600 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3 --batch 32 2>&1 | grep "^\[" | tee -a "$OUT/summary.txt"
timeout 900 python tools/bench_configs.py long 2>&1 | tail -1 | tee "$OUT/long_utterance_750.json" | cut -c1-600
timeout 600 python tools/bench_configs.py codec_only 2>&1 | tail -1 | tee "$OUT/config2_codec_only.json" | cut -c1-900
# the unrolled step's scheduling hint: "1 MFMA + up to N others", N = 0 (no hints) / 2 (product) / 3 / 4 -- the A/B tool per build variant
for v in "" _ring_sgb0 _ring_sgb3 _ring_sgb4; do
  QTTS_LIBRARY="$PWD/qwen3-tts_amd/libqtts$v.so" timeout 600 python tools/bench_gemm_ring.py --skip-linear --screen 0 > "$OUT/bench_ring$v.log" 2>&1
  echo "== libqtts$v.so" | tee -a "$OUT/summary.txt"; grep -E "^(C768 conv7 d1|C384 conv7 d3|tconv 768|32 x 4)" "$OUT/bench_ring$v.log" | head -7 | cut -c1-110 | tee -a "$OUT/summary.txt"
done
