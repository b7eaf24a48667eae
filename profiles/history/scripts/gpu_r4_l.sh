#!/usr/bin/env bash
# Round 4, GPU call 13: the code predictor's attention + o-projection as one launch (cp_attn_o_kernel) -- its GPU test (run-to-run identical,
# against the two launches), the bf16 pin at the metric config, then the frame step A/B in alternating processes, a kernel trace, the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4l
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_fused 600 python -m pytest tests -q -m gpu -x -s -k "fused_attention or bf16_mode_pinned or tiny_greedy or talker_bf16_mode_tracks"
for i in 1 2; do
  QTTS_CP_ATTN_O=1 run frame_fused_$i 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
  QTTS_CP_ATTN_O=0 run frame_plain_$i 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
done
prof trace_frame --kernel-trace --stats -d "$PWD/$OUT/tr1" -o perf -- python "$PWD/tools/perf_frame.py" --model 1.7b --frames 30 --talker --reps 1
TDB=$(find "$OUT/tr1" -name "*.db" | head -1)
[ -n "$TDB" ] && python tools/rocpd_stats.py "$TDB" --out "$OUT/kernel_trace_frame_fused.md" > /dev/null 2>&1
rm -rf "$OUT/tr1"
run bench 600 python bench.py --steps 10 --warmup 3 --no-parity-mode
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
grep -h "sampling\|greedy" "$OUT"/frame_*.log | cut -c1-170
head -12 "$OUT/kernel_trace_frame_fused.md" | cut -c1-160
cat "$OUT/summary.txt"
