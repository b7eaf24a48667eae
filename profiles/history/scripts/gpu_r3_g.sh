#!/usr/bin/env bash
# Round 3, GPU call 7: in-kernel phase timestamps of the fused residual unit (tstamp variant).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3g
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
timeout 300 python tools/ts_codec.py --json "$OUT/ts_codec.json" > "$OUT/ts_codec.log" 2>&1; echo "ts_codec rc=$?"
tail -30 "$OUT/ts_codec.log"
QTTS_RESUNIT_TM=4 timeout 300 python tools/ts_codec.py > "$OUT/ts_codec_tm4.log" 2>&1; echo "ts_codec_tm4 rc=$?"
tail -8 "$OUT/ts_codec_tm4.log"
