#!/usr/bin/env bash
# Round 2, GPU call 17: attn_tk at longer KV (in-kernel timestamps), frame time vs KV length.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2k
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; grep -E "attn_tk|attn_cp|sampler|pitch" "$OUT/$name.log" | cut -c1-200 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run ts_p0 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_p0.json"
run ts_p60 200 python tools/ts_frame.py --model 1.7b --frames 12 --prompt 60 --json "$OUT/ts_p60.json"
run ts_p100 200 python tools/ts_frame.py --model 1.7b --frames 12 --prompt 100 --json "$OUT/ts_p100.json"
run ts_p200 200 python tools/ts_frame.py --model 1.7b --frames 12 --prompt 200 --json "$OUT/ts_p200.json"
cat "$OUT/summary.txt"
