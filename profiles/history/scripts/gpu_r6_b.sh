#!/usr/bin/env bash
# Round 6, call b: cp_layer_kernel's first hardware run -- bit-identity against the two fused launches, the fused / contention / fp32-instantiation
# tests through it, in-process A/B of the frame step (layer launch vs two launches; DMA timing; first-read pause of the hidden rows).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6b
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-900 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=8 run pytest_layer 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "whole_layer_launch"
TAILN=14 run ab 700 python tools/ab_inproc.py --frames 40 --reps 3 --only default cp_layer_off layer_gu_late layer_h8 layer_h24 layer_h32
TAILN=8 run pytest_fused 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "fused or contention or bf16_parity or tiny_greedy"
cat "$OUT/summary.txt"
