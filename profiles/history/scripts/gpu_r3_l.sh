#!/usr/bin/env bash
# Round 3, GPU call 12: the hand-off probe again (barrier in the XCD-hierarchical form, phase clocks, and the third variant:
# launches + an operator-prefetch branch in the same graph).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3l
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
timeout 120 python tools/persist_probe.py > "$OUT/persist_probe.log" 2>&1; echo "probe rc=$?"; tail -3 "$OUT/persist_probe.log"
timeout 200 python tools/cold_chain.py > "$OUT/cold_chain.log" 2>&1; echo "cold rc=$?"; cat "$OUT/cold_chain.log"
