#!/usr/bin/env bash
# Round 6, call x: the SQ counters of call n again on the ring kernel (LDS bank conflicts, LDS active, wait classes, MFMA busy), codec at 8 x 10 s,
# per-dispatch; and the per-dispatch kernel trace at 8 x 10 s and 1 x 10 s.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6x
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
for B in 8 1; do
  prof trace_b$B --kernel-trace -d "$PWD/$OUT/tr$B" -o perf -- python "$PWD/tools/perf_frame.py" --codec --reps 2 --batch $B
  DB=$(find "$OUT/tr$B" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_dispatches.py "$DB" "gemm_ring|gemm_dma|resunit|gemm_tap2" --out "$OUT/dispatches_b$B.md" > /dev/null 2>&1; rm -rf "$OUT/tr$B"
done
prof pmc_a --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d "$PWD/$OUT/pmca" -o pmc -- python "$PWD/tools/perf_frame.py" --codec --reps 1 --batch 8
DB=$(find "$OUT/pmca" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_dispatches.py "$DB" "gemm_ring|gemm_dma|resunit" --pmc --out "$OUT/pmc_sq_b8.md" > /dev/null 2>&1; rm -rf "$OUT/pmca"
cat "$OUT/summary.txt"
