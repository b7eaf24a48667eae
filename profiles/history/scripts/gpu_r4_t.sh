#!/usr/bin/env bash
# Round 4, GPU call 21: the fused launch's GPU test with its batch-3 part, and the bench line with `roofline.fused_cp_launch` on it.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4t
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-700 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_fused 600 python -m pytest tests -q -m gpu -x -s -k "fused_attention"
run bench 900 python bench.py --steps 20 --warmup 5
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
grep -h "cp_attn_o\|batch 3" "$OUT"/pytest_fused.log | cut -c1-250
cat "$OUT/summary.txt"
