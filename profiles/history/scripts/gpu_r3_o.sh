#!/usr/bin/env bash
# Round 3, GPU call 15: small-grid GEMM per tile incl. 256-wide k-steps (and the wide kernel on grids above 1024 tiles); the frame /
# prefill times and the codec decode with the tile chooser.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3o
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
timeout 300 python tools/bench_gemm_small.py > "$OUT/gemm_small.log" 2>&1; echo "gemm rc=$?"; cat "$OUT/gemm_small.log"
QTTS_GEMM_WIDE_MAX=4096 timeout 300 python tools/bench_gemm_small.py > "$OUT/gemm_small_widemax.log" 2>&1; echo "gemm rc=$?"; grep "b32" "$OUT/gemm_small_widemax.log"
timeout 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker > "$OUT/perf_frame.log" 2>&1; tail -3 "$OUT/perf_frame.log"
QTTS_GEMM_NARROW=0 timeout 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker > "$OUT/perf_frame_r2rule.log" 2>&1; tail -3 "$OUT/perf_frame_r2rule.log"
timeout 120 python tools/bench_configs.py codec_only > "$OUT/codec.log" 2>&1
echo "codec: $(grep -o '"dtype": "bf16", "batch": [18], "ms_p50": [0-9.]*' "$OUT/codec.log" | tr '\n' ' ')"
timeout 200 python tools/bench_configs.py first_packet > "$OUT/first_packet.log" 2>&1; tail -1 "$OUT/first_packet.log"
