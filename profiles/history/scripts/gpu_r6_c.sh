#!/usr/bin/env bash
# Round 6, call c: in-kernel timeline of the layer launch (tstamp build) against the two fused launches -- where do the +3 us per layer go?
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6c
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-400 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=30 run ts_layer 400 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_layer.json"
QTTS_CP_LAYER=0 TAILN=30 run ts_two 400 env QTTS_CP_LAYER=0 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_two.json"
cat "$OUT/summary.txt"
