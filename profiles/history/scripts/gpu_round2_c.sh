#!/usr/bin/env bash
# Round 2, GPU call 5: attn_tk + gemm_tap2 on hardware (full suite), frame timing, fs-floor A/B, codec config 2 with and
# without the bf16-activation path, long utterance, MFMA-busy counters of the codec GEMMs.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2c
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n 4 "$OUT/$name.log" | cut -c1-300 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run pytest_gpu 900 python -m pytest tests -q -m gpu -s
run perf_frame 240 python tools/perf_frame.py --model 1.7b --frames 60 --talker --prof
run ab_fs 400 python tools/ab_inproc.py --frames 40 --reps 2
run codec_fast16 300 python tools/bench_configs.py codec_only --trials 10
run codec_old 300 env QTTS_CODEC_FAST16=0 python tools/bench_configs.py codec_only --trials 10
run long 420 python tools/bench_configs.py long --frames 750
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d "$OLDPWD/$OUT/pmc" -o pmc -- python "$OLDPWD/tools/perf_frame.py" --codec --reps 1 --batch 8 > "$OLDPWD/$OUT/pmc.log" 2>&1 ); echo "pmc rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/pmc" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_mfma_codec.md" > /dev/null 2>&1; rm -rf "$OUT/pmc"
cat "$OUT/summary.txt"
