#!/usr/bin/env bash
# Round 6, call f: layer launch v4 (reducers request their gate|up block early, the others behind the attention stage, optionally paced; one layer engine
# per device): A/B, timeline, contention test, the fp32 instantiation through the reference's goldens.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6f
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-500 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=22 run ab 900 python tools/ab_inproc.py --frames 40 --reps 4 --only default cp_layer_off layer_pace4 layer_pace8 layer_hid1 layer_gu_entry
TAILN=6 run ts_layer 400 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_layer.json"
TAILN=6 run ts_layer_pace8 400 env QTTS_CP_LAYER_GU_PACE=8 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_layer_pace8.json"
TAILN=6 run pytest_layer 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "whole_layer_launch or contention"
TAILN=10 run pytest_f32 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "fp32_instantiations"
cat "$OUT/summary.txt"
