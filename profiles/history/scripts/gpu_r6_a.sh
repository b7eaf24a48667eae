#!/usr/bin/env bash
# Round 6, call a: the new sampled-path test at the benchmarked shape (2000 seeds), the fused-launch tests on the new tree, frame-step baseline for this round's A/Bs.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6a
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-900 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=12 run pytest_sampler 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "sampled_path or sampler_distribution or code_predictor_sampling"
TAILN=6 run pytest_fused 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "fused or tiny_greedy or contention"
TAILN=12 run ab 600 python tools/ab_inproc.py --frames 40 --reps 3 --only default cp_mlp_off
cat "$OUT/summary.txt"
