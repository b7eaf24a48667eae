#!/usr/bin/env bash
# Round 6, call t: ring kernel v4 with the static last slab, default for the prefill's Linears on grids >= 128 tiles: the WHOLE GPU suite, the A/B table,
# the bench line (configs leg: config 2 / 4 / 5), MFMA-busy counters of the codec at 8 x 10 s.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6t
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
: > "$OUT/summary.txt"
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-400 | sed "s/^/    /"; }
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
TAILN=40 run bench_ring 900 python tools/bench_gemm_ring.py --screen 2
TAILN=4 run pytest_gpu 1700 python -m pytest tests -q -m gpu -s
run bench 900 python bench.py
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
prof pmc_mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d "$PWD/$OUT/pmc2" -o pmc -- python "$PWD/tools/perf_frame.py" --codec --reps 1 --batch 8
DB=$(find "$OUT/pmc2" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_mfma_codec_raw.md" > /dev/null 2>&1; rm -rf "$OUT/pmc2"
prof trace_codec --kernel-trace --stats -d "$PWD/$OUT/tr2" -o perf -- python "$PWD/tools/perf_frame.py" --codec --reps 3 --batch 8
DB=$(find "$OUT/tr2" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/codec_kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/tr2"
cat "$OUT/summary.txt"
