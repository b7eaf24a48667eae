#!/usr/bin/env bash
# Round 6, call k: cp_mlp32 GPU parity test (+ the split-K test with the fused MLP off), clone-shard with engines that share a device keeping the
# decode GEMMs (talker.py: shared_device), rocprofv3 kernel trace of the batch-32 frame step with and without the round's two batch-32 kernels.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6k
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_b32 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "fused_mlp_launch_at_batch_32 or split_k_decode"
for i in 1 2; do TAILN=2 run config5_$i 600 python bench.py --workload clone-shard --steps 1 --warmup 1 --no-cpu-baseline; done
prof trace_b32 --kernel-trace --stats -d "$PWD/$OUT/tr1" -o perf -- python "$PWD/tools/perf_frame.py" --model 1.7b --batch 32 --frames 24 --talker --reps 2
DB=$(find "$OUT/tr1" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace_frame_b32.md" > /dev/null 2>&1; rm -rf "$OUT/tr1"
prof trace_b32_off --kernel-trace --stats -d "$PWD/$OUT/tr2" -o perf -- env QTTS_CP_MLP32=0 QTTS_SKINNY_KS=0 python "$PWD/tools/perf_frame.py" --model 1.7b --batch 32 --frames 24 --talker --reps 2
DB=$(find "$OUT/tr2" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace_frame_b32_round5.md" > /dev/null 2>&1; rm -rf "$OUT/tr2"
head -14 "$OUT/kernel_trace_frame_b32.md" | cut -c1-160
cat "$OUT/summary.txt"
