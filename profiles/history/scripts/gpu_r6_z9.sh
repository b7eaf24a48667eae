#!/usr/bin/env bash
# Round 6, call z9: 32-row tiles of the wide-K GEMM for launches of <= 128 rows (codec transformer at 1 x 10 s / 32 x 4 frames, one-utterance prefill):
# per-tile microbenchmark, config 2 and config 4 A/B against the forced round-5 tile, the codec / prefill / talker-golden GPU tests on the new chooser.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6z9
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
timeout 400 python tools/bench_gemm_small.py --small > "$OUT/gemm_small_tiles.txt" 2>&1; echo "tiles rc=$?"; cat "$OUT/gemm_small_tiles.txt" | cut -c1-220
for v in new old new old; do
  if [ $v = old ]; then export QTTS_GEMM_WIDE_TILE=64064256; else unset QTTS_GEMM_WIDE_TILE; fi
  echo "== config 2 ($v)"; timeout 300 python tools/bench_configs.py codec_only --batch 1 --trials 30 2>&1 | tail -1 | cut -c1-400 | tee -a "$OUT/config2_ab_$v.json"
done
unset QTTS_GEMM_WIDE_TILE
echo "== config 4 (new)"; timeout 400 python tools/bench_configs.py first_packet --trials 20 2>&1 | tail -1 | cut -c1-300 | tee "$OUT/config4_new.json"
timeout 1200 python -m pytest tests -q -m gpu -k "codec or prefill or prompt_assembly or wrapper or 17b_ragged or 06b_one or stream" > "$OUT/pytest_subset.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_subset.log"
