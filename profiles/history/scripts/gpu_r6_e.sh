#!/usr/bin/env bash
# Round 6, call e: who makes the layer launch give up (tools/diag_layer_pair.py), the third version of the kernel (gate|up tiles LDS -> registers during
# the hidden read) in-process against the two launches, its timeline.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6e
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-500 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=14 run diag 900 python tools/diag_layer_pair.py
TAILN=4 run pytest_layer 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "whole_layer_launch"
TAILN=20 run ab 900 python tools/ab_inproc.py --frames 40 --reps 4 --only default cp_layer_off layer_hid0 layer_h8 layer_h24
TAILN=8 run ts_layer 400 env QTTS_CP_LAYER_HID_MODE=0 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_layer_hid0.json"
cat "$OUT/summary.txt"
