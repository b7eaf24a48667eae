#!/usr/bin/env bash
# Round 6, call d: cp_layer_kernel second version (DMA behind the attention stage, sentinel polling of the hidden rows, rotating regions, LDS aliasing):
# bit-identity, contention, in-process A/B of the modes, timeline.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6d
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-400 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_layer 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "whole_layer_launch or contention"
TAILN=24 run ab 900 python tools/ab_inproc.py --frames 40 --reps 3 --only default cp_layer_off layer_gu_entry layer_hid0 layer_hid2 layer_h8 layer_h24
TAILN=8 run ts_layer 400 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_layer.json"
grep -E "cp_layer|cp_attn_o|cp_mlp" "$OUT/ts_layer.log" | cut -c1-300
QTTS_CP_LAYER_HID_MODE=2 TAILN=8 run ts_layer_hid2 400 env QTTS_CP_LAYER_HID_MODE=2 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_layer_hid2.json"
grep -E "cp_layer|cp_attn_o|cp_mlp" "$OUT/ts_layer_hid2.log" | cut -c1-300
cat "$OUT/summary.txt"
