#!/usr/bin/env bash
# Round 2, GPU calls 21-27: codec GEMM experiments -- timing (A/B by environment) + parity.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2m
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-2} "$OUT/$name.log" | cut -c1-250 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run perf_codec 300 python tools/perf_frame.py --model 1.7b --codec --codec-frames 125 --reps 7
run perf_codec_res32 300 env QTTS_CODEC_RES16=0 python tools/perf_frame.py --model 1.7b --codec --codec-frames 125 --reps 7
run perf_codec_b1 300 python tools/perf_frame.py --model 1.7b --codec --codec-frames 125 --reps 7 --batch 1
run perf_codec_f32 300 python tools/perf_frame.py --model 1.7b --codec --codec-frames 125 --reps 3 --codec-dtype f32
TAILN=8 run pytest_codec 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -s -k "codec or encoder or speaker or clone"
grep -h "relative RMS\|rel RMS\|rel. RMS\|bf16" "$OUT/pytest_codec.log" | head -8
cat "$OUT/summary.txt"
