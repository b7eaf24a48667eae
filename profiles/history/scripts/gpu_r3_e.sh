#!/usr/bin/env bash
# Round 3, GPU call 5: resunit.hip at two workgroups per CU for C = 96 (128-row tiles) vs one (QTTS_RESUNIT_TM=4) -- codec parity,
# config 2, kernel trace + MFMA-busy counters.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3e
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-900 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=8 run pytest_codec 600 python -m pytest tests -q -m gpu -s -k "codec or wrapper or smoke"
run codec_tm2 300 python tools/bench_configs.py codec_only --trials 10
QTTS_RESUNIT_TM=4 run codec_tm4 300 python tools/bench_configs.py codec_only --trials 10
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o perf -- python "$OLDPWD/tools/perf_frame.py" --codec --reps 3 --batch 8 > "$OLDPWD/$OUT/rocprof_codec.log" 2>&1 ); echo "rocprof_codec rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/codec_kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof"
head -12 "$OUT/codec_kernel_trace.md"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d "$OLDPWD/$OUT/pmc" -o pmc -- python "$OLDPWD/tools/perf_frame.py" --codec --reps 1 --batch 8 > "$OLDPWD/$OUT/pmc_mfma.log" 2>&1 ); echo "pmc_mfma rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/pmc" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_mfma_codec.md" > /dev/null 2>&1; rm -rf "$OUT/pmc"
grep -E "resunit|gemm_tap2" "$OUT/pmc_mfma_codec.md" | head -12
for f in codec_tm2 codec_tm4; do grep -h '^{' "$OUT/$f.log" | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$f', [(r['dtype'],r['batch'],r['ms_p50']) for r in j['runs']])"; done
cat "$OUT/summary.txt"
