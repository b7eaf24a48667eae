#!/usr/bin/env bash
# Round 2, GPU call 10: skinny8_kernel (tile pairs, unconditional whole-line requests) and the bit-select sampler.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2h
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-6} "$OUT/$name.log" | cut -c1-250 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run pytest_subset 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "tiny or talker_06b_one or talker_17b_ragged or pair_kernel"
run ab 100 true
TAILN=22 run ts_graph 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_graph.json"
run perf_frame 240 python tools/perf_frame.py --model 1.7b --frames 60 --talker --prof
cat "$OUT/summary.txt"
