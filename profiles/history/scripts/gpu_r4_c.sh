#!/usr/bin/env bash
# Round 4, GPU call 3: the whole GPU suite on the round's tree (fp32 batch <= 8 kernel, hygiene, top_p = 0, KV append guard), the bench line
# with the new legs (parity_mode.roofline, api_e2e), BASELINE config 5 through bench.py with the round-4 defaults (waves of 32, two
# engines per GPU, device gather) against round 3's shape (waves of 8, one engine), and the fp32 wave-count A/Bs at K = 1024 / 3072.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4c
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-1200 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_gpu 900 python -m pytest tests -q -m gpu
run smoke 200 python __graft_entry__.py --smoke
run bench 600 python bench.py --steps 10 --warmup 3
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
run clone_r4 600 python bench.py --workload clone-shard --steps 1 --warmup 1
grep -h '^{' "$OUT/clone_r4.log" > "$OUT/clone_r4.json"
run clone_b32_e1 400 python bench.py --workload clone-shard --steps 1 --warmup 1 --engines 1
run clone_r3 400 python bench.py --workload clone-shard --steps 1 --warmup 1 --engines 1 --batch 8
run f32_dflt 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --talker-dtype f32 --reps 2 --prof
QTTS_SKINNY8F_NW=16 run f32_nw16 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --talker-dtype f32 --reps 2
QTTS_SKINNY8F_NW=8 run f32_nw8 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --talker-dtype f32 --reps 2
grep -h "greedy\|sampling" "$OUT"/f32_*.log | cut -c1-200
grep -h "stack\|decode GEMM" "$OUT"/f32_dflt.log | cut -c1-200
cat "$OUT/summary.txt"
