#!/usr/bin/env bash
# Round 4, GPU call 6: (1) batch 17..32 decode GEMM with whole-line x requests over k-tile pairs (PAIRX) -- bf16 batch tests, batch-32 frame
# and first packet on / off; (2) VERDICT r3 item 6a as a measurement: the talker's decode attention split over 2-4 workgroups per
# (sequence, kv head) at SHORT lengths (existing split-KV + merge kernel, thresholds lowered by environment): kernel trace on / off.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4f
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-1200 | sed "s/^/    /"; }
trace() { local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_$name" -o perf -- python "$OLDPWD/tools/perf_frame.py" "$@" > "$OLDPWD/$OUT/rocprof_$name.log" 2>&1 ); echo "rocprof_$name rc=$?" | tee -a "$OUT/summary.txt"
  DB=$(find "$OUT/prof_$name" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace_$name.md" > /dev/null 2>&1; rm -rf "$OUT/prof_$name"
  grep "sampling\|greedy" "$OUT/rocprof_$name.log" | cut -c1-160; grep "attn_\|skinny" "$OUT/kernel_trace_$name.md" | head -8 | cut -c1-150; }
: > "$OUT/summary.txt"
TAILN=5 run pytest_b32 600 python -m pytest tests -q -m gpu -x -k "large_batch or batch32 or bf16_mode or pair_kernel"
run b32_pairx 200 python tools/perf_frame.py --model 1.7b --frames 40 --talker --batch 32 --reps 3
QTTS_SKINNY2_PAIRX=0 run b32_frag 200 python tools/perf_frame.py --model 1.7b --frames 40 --talker --batch 32 --reps 3
run b32_pairx2 200 python tools/perf_frame.py --model 1.7b --frames 40 --talker --batch 32 --reps 3
QTTS_SKINNY2_PAIRX=0 run b32_frag2 200 python tools/perf_frame.py --model 1.7b --frames 40 --talker --batch 32 --reps 3
run fp_pairx 200 python tools/bench_configs.py first_packet --trials 10
QTTS_SKINNY2_PAIRX=0 run fp_frag 200 python tools/bench_configs.py first_packet --trials 10
trace attn_default --model 1.7b --frames 40 --talker --reps 2
QTTS_ATTN_SPLIT_FROM=1 QTTS_ATTN_SPLIT_KEYS=64 QTTS_ATTN_NSPLIT=4 trace attn_split4 --model 1.7b --frames 40 --talker --reps 2
grep -h "greedy\|sampling" "$OUT"/b32_*.log | cut -c1-170
cat "$OUT/summary.txt"
