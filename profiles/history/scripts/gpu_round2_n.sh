#!/usr/bin/env bash
# Round 2, GPU call 26: in-kernel timestamps of the codec's tap-reuse GEMM.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2n
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
timeout 300 python tools/ts_codec.py --json "$OUT/ts_codec.json" > "$OUT/ts_codec.log" 2>&1; echo "rc=$?"; tail -30 "$OUT/ts_codec.log" | cut -c1-200
