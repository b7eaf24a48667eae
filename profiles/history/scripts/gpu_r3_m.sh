#!/usr/bin/env bash
# Round 3, GPU call 13: where is a decode-GEMM launch of the frame step longer than the same launch in a chain of its own?
# (a) chains with the engine's strip widths, operator resident vs streamed; (b) the same chains on the tstamp variant.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3m
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
timeout 200 python tools/cold_chain.py > "$OUT/cold_chain.log" 2>&1; echo "cold rc=$?"; cat "$OUT/cold_chain.log"

