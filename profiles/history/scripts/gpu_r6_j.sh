#!/usr/bin/env bash
# Round 6, call j: the code predictor's MLP as one launch at batch 9..32 (cp_mlp32_kernel): GPU parity test, in-process A/B at batch 32 (with its
# first-read pauses), configs 4 / 5 with and without, the batch-8 tests of the fused launches (nothing of theirs changed: the engine's admission did).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6j
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_mlp32 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "fused_mlp_launch_at_batch_32 or split_k_decode"
TAILN=12 run ab_b32 900 python tools/ab_inproc.py --batch 32 --frames 40 --reps 3 --only default mlp32_off mlp32_off_ks_off mlp32_b16 mlp32_b32 mlp32_b40 mlp32_c32 mlp32_c40
cp gpurun_out/ab_inproc_b32.json "$OUT/" 2>/dev/null
TAILN=3 run config4 600 python tools/bench_configs.py first_packet
TAILN=3 run config4_off 600 env QTTS_CP_MLP32=0 QTTS_SKINNY_KS=0 python tools/bench_configs.py first_packet
for i in 1 2; do TAILN=2 run config5_$i 600 python bench.py --workload clone-shard --steps 1 --warmup 1 --no-cpu-baseline; done
TAILN=2 run config5_off 600 env QTTS_CP_MLP32=0 QTTS_SKINNY_KS=0 python bench.py --workload clone-shard --steps 1 --warmup 1 --no-cpu-baseline
TAILN=6 run pytest_b8 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "contention or whole_layer_launch or fused_attention_o or fused_mlp_equals"
cat "$OUT/summary.txt"
