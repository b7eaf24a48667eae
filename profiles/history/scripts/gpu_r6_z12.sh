#!/usr/bin/env bash
# Round 6, call z12: single-stream first packet (batch 1 and 2) with the 32-row tiles against the forced round-5 tile.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6z12
mkdir -p "$OUT"
for b in 1 2; do for v in new old new old; do
  if [ $v = old ]; then export QTTS_GEMM_WIDE_TILE=64064256; else unset QTTS_GEMM_WIDE_TILE; fi
  echo -n "batch $b $v: "; timeout 300 python tools/bench_configs.py first_packet --batch $b --trials 30 2>&1 | tail -1 | cut -c1-330 | tee -a "$OUT/first_packet_b${b}_$v.json"
done; done
