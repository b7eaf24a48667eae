#!/usr/bin/env bash
# Round 2, GPU call 9: in-kernel phase timestamps of the frame step (build variant `tstamp`, tools/ts_frame.py).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2g
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n 25 "$OUT/$name.log" | cut -c1-220 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run ts_graph 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_graph.json"
run ts_eager 200 python tools/ts_frame.py --model 1.7b --frames 12 --no-graph --json "$OUT/ts_eager.json"
run ts_graph_long 200 python tools/ts_frame.py --model 1.7b --frames 12 --prompt 200 --json "$OUT/ts_graph_long.json"
cat "$OUT/summary.txt"
