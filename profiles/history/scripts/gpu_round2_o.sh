#!/usr/bin/env bash
# Round 2, GPU call 35: MFMA-busy counters of the bf16 codec decode on the round's last build.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2o
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d "$OLDPWD/$OUT/pmc2" -o pmc -- python "$OLDPWD/tools/perf_frame.py" --codec --reps 1 --batch 8 > "$OLDPWD/$OUT/pmc_mfma.log" 2>&1 ); echo "pmc_mfma rc=$?"
DB=$(find "$OUT/pmc2" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_mfma_codec.md" > /dev/null 2>&1; rm -rf "$OUT/pmc2"
grep "gemm_tap2" "$OUT/pmc_mfma_codec.md" | cut -c1-160
