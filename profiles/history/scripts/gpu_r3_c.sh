#!/usr/bin/env bash
# Round 3, GPU call 3: the fused residual-unit kernel of the codec decoder (resunit.hip, C = 96 / 192) -- codec parity tests,
# BASELINE config 2 (codec decode-only) fused vs the two-launch path (QTTS_CODEC_FUSED=0), kernel trace and MFMA-busy counters of
# the bf16 codec decode, bench.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3c
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-900 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=8 run pytest_codec 600 python -m pytest tests -q -m gpu -s -k "codec or wrapper or smoke"
run codec_fused 300 python tools/bench_configs.py codec_only --trials 10
QTTS_CODEC_FUSED=0 run codec_unfused 300 python tools/bench_configs.py codec_only --trials 10
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o perf -- python "$OLDPWD/tools/perf_frame.py" --codec --reps 3 --batch 8 > "$OLDPWD/$OUT/rocprof_codec.log" 2>&1 ); echo "rocprof_codec rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/codec_kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof"
head -14 "$OUT/codec_kernel_trace.md"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d "$OLDPWD/$OUT/pmc" -o pmc -- python "$OLDPWD/tools/perf_frame.py" --codec --reps 1 --batch 8 > "$OLDPWD/$OUT/pmc_mfma.log" 2>&1 ); echo "pmc_mfma rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/pmc" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_mfma_codec.md" > /dev/null 2>&1; rm -rf "$OUT/pmc"
head -30 "$OUT/pmc_mfma_codec.md"
run bench 420 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
cat "$OUT/summary.txt"
