#!/usr/bin/env bash
# Round 6, call i: the split-K reducer's first-read pause (the producers' stores take ~1.5 us to land: a first read that leaves too early costs a
# whole second round of 6 / 28 sc1 reads), clone-shard three times (the legacy-stream copy fix), GPU parity test with its re-measured bar.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6i
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=12 run ab_b32 900 python tools/ab_inproc.py --batch 32 --frames 40 --reps 3 --only default ks_off ks_mink6144 ks_pause32 ks_pause48 ks_pause64 ks_mink6144_p32 ks_mink6144_p48
cp gpurun_out/ab_inproc_b32.json "$OUT/" 2>/dev/null
TAILN=6 run ts_b32_p48 400 env QTTS_SKINNY_KS_PAUSE=48 python tools/ts_frame.py --model 1.7b --batch 32 --frames 8 --json "$OUT/ts_b32_p48.json"
TAILN=4 run pytest_ks 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "split_k_decode"
for i in 1 2 3; do TAILN=2 run config5_$i 600 python bench.py --workload clone-shard --steps 1 --warmup 1 --no-cpu-baseline; done
cat "$OUT/summary.txt"
