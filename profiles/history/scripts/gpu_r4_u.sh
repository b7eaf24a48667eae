#!/usr/bin/env bash
# Round 4, GPU call 22: the fused launch at batch sizes below 8 (debug).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4u
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
timeout 500 python tools/debug_cp_fused.py > "$OUT/debug.log" 2>&1; echo "rc=$?"
grep -v "amdgpu.ids" "$OUT/debug.log" | tail -60 | cut -c1-300
