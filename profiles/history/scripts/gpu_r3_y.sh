#!/usr/bin/env bash
# Round 3, GPU call 26: the PMC passes over the frame step's kernels re-taken on the round's LAST tree (gate|up now runs as
# skinny8_kernel<1, 16, 4, true, 8> at 768 workgroups): FETCH_SIZE, and the SQ wave / wait counters.  Counters only, kernel trace only.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3y
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?"; }
prof pmc_fetch --kernel-trace --pmc FETCH_SIZE -d "$PWD/$OUT/pmc1" -o pmc -- python "$PWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph
DB=$(find "$OUT/pmc1" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_fetch_size.md" > /dev/null 2>&1; rm -rf "$OUT/pmc1"
grep skinny8 "$OUT/pmc_fetch_size.md" | cut -c1-120
prof pmc_sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d "$PWD/$OUT/pmc2" -o pmc -- python "$PWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph
DB=$(find "$OUT/pmc2" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_sq_frame_kernels.md" > /dev/null 2>&1; rm -rf "$OUT/pmc2"
grep -c skinny8 "$OUT/pmc_sq_frame_kernels.md"
