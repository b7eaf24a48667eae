#!/usr/bin/env bash
# Round 3, GPU call 14: small-grid GEMM tiles (64-row tiles, bf16 activations) per shape; codec decode with the bf16 hand-over to the
# final convolution (QTTS_CODEC_FINAL16 A/B); codec parity tests on the new tail.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3n
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
timeout 300 python tools/bench_gemm_small.py > "$OUT/gemm_small.log" 2>&1; echo "gemm rc=$?"; cat "$OUT/gemm_small.log"
for i in 1 2; do
  timeout 120 python tools/bench_configs.py codec_only > "$OUT/codec_final16_$i.log" 2>&1
  QTTS_CODEC_FINAL16=0 timeout 120 python tools/bench_configs.py codec_only > "$OUT/codec_final32_$i.log" 2>&1
done
for f in "$OUT"/codec_final*.log; do echo "$f: $(grep -o '"dtype": "bf16", "batch": [18], "ms_p50": [0-9.]*' "$f" | tr '\n' ' ')"; done
timeout 600 python -m pytest tests -q -m gpu -x -k "codec" 2>&1 | tail -5
