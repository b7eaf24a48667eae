#!/usr/bin/env bash
# Round 6, call h: split-K decode GEMM at batch 32, second build (the polling re-read pinned inside its loop): GPU parity test, in-process A/B at
# batch 32, in-kernel timeline at batch 32, configs 4 / 5 with and without.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6h
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_ks 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "split_k_decode"
TAILN=10 run ab_b32 900 python tools/ab_inproc.py --batch 32 --frames 40 --reps 3 --only default ks_off ks_mink2048 ks_mink6144 ks_pause0 ks_pause16
cp gpurun_out/ab_inproc_b32.json "$OUT/" 2>/dev/null
TAILN=30 run ts_b32 400 python tools/ts_frame.py --model 1.7b --batch 32 --frames 8 --json "$OUT/ts_b32.json"
TAILN=30 run ts_b32_mink2048 400 env QTTS_SKINNY_KS_MINK=2048 python tools/ts_frame.py --model 1.7b --batch 32 --frames 8 --json "$OUT/ts_b32_mink2048.json"
TAILN=24 run ts_b32_off 400 env QTTS_SKINNY_KS=0 python tools/ts_frame.py --model 1.7b --batch 32 --frames 8 --json "$OUT/ts_b32_off.json"
TAILN=3 run config4 600 python tools/bench_configs.py first_packet
TAILN=3 run config4_off 600 env QTTS_SKINNY_KS=0 python tools/bench_configs.py first_packet
TAILN=3 run config5 900 python bench.py --workload clone-shard --steps 1 --warmup 1 --no-cpu-baseline
TAILN=3 run config5_off 900 env QTTS_SKINNY_KS=0 python bench.py --workload clone-shard --steps 1 --warmup 1 --no-cpu-baseline
cat "$OUT/summary.txt"
