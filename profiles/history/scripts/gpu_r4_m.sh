#!/usr/bin/env bash
# Round 4, GPU call 14: where the fused attention + o-projection launch spends its time -- in-kernel timestamps (tstamp build) of the frame step
# with the fused launch and with the two launches.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4m
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
export QTTS_LIBRARY_OK=1
TAILN=30 QTTS_CP_ATTN_O=1 run ts_fused 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_fused.json"
TAILN=30 QTTS_CP_ATTN_O=0 run ts_plain 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_plain.json"
cat "$OUT/summary.txt"
