#!/usr/bin/env bash
# Round 2, GPU call 15: unconditional k-loop loads in the tile GEMMs (codec, prefill).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2i
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-6} "$OUT/$name.log" | cut -c1-250 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run pytest_codec 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "codec or encoder or speaker or prompt"
run perf_codec 300 python tools/perf_frame.py --model 1.7b --codec --codec-frames 125 --reps 5
run perf_codec_f32 300 python tools/perf_frame.py --model 1.7b --codec --codec-frames 125 --reps 3 --codec-dtype f32
run perf_frame 240 python tools/perf_frame.py --model 1.7b --frames 60 --talker
cat "$OUT/summary.txt"
