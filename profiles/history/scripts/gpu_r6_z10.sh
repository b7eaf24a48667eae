#!/usr/bin/env bash
# Round 6, call z10: config 2 (bf16) A/B of the 32-row tiles against the forced round-5 tile, two rounds each, whole lines kept.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6z10
mkdir -p "$OUT"
for v in new old new old; do
  if [ $v = old ]; then export QTTS_GEMM_WIDE_TILE=64064256; else unset QTTS_GEMM_WIDE_TILE; fi
  timeout 300 python tools/bench_configs.py codec_only --batch 1 --trials 30 2>&1 | tail -1 >> "$OUT/config2_ab_$v.json"
done
python - <<'P'
import json
for v in ("new", "old"):
    for l in open(f"gpurun_out/r6z10/config2_ab_{v}.json"):
        d = json.loads(l)
        print(v, [(r["dtype"], r["batch"], r["ms_p50"], r["ms_min"]) for r in d["runs"]])
P
