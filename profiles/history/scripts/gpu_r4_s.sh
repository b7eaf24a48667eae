#!/usr/bin/env bash
# Round 4, GPU call 20: the tree with the fused code-predictor launch on by default -- frame step A/B once more (build 7), timeline, two engines
# per GPU at batch 8 (two fused launches resident at the same time), the stamped counters / trace for this tree's decode GEMM, the whole GPU
# suite, smoke(), the bench line with every leg.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4s
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-700 | sed "s/^/    /"; }
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
run frame_front_1 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
QTTS_CP_ATTN_O=0 run frame_plain_1 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
run frame_front_2 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
QTTS_CP_FRONT=0 run frame_attno_1 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
TAILN=30 QTTS_LIBRARY_OK=1 run ts_front 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_front.json"
TAILN=2 run clone_b8_e2 300 python bench.py --workload clone-shard --batch 8 --engines 2 --requests 64 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-api-e2e
TAILN=2 QTTS_CP_ATTN_O=0 run clone_b8_e2_plain 300 python bench.py --workload clone-shard --batch 8 --engines 2 --requests 64 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-api-e2e
prof pmc_fetch --kernel-trace --pmc FETCH_SIZE -d "$PWD/$OUT/pmc1" -o pmc -- python "$PWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph
prof trace_bench --kernel-trace --stats -d "$PWD/$OUT/tr1" -o perf -- python "$PWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --no-api-e2e --no-roofline
FDB=$(find "$OUT/pmc1" -name "*.db" | head -1); TDB=$(find "$OUT/tr1" -name "*.db" | head -1)
[ -n "$FDB" ] && python tools/rocpd_pmc.py "$FDB" --out "$OUT/pmc_fetch_size.md" > /dev/null 2>&1
[ -n "$TDB" ] && python tools/rocpd_stats.py "$TDB" --out "$OUT/rocprofv3_kernel_trace_bench.md" > /dev/null 2>&1
[ -n "$FDB" ] && [ -n "$TDB" ] && python tools/pmc_traffic.py --fetch-db "$FDB" --trace-db "$TDB" --source "profiles/r04_pmc_fetch_size.md + profiles/r04_rocprofv3_kernel_trace_bench.md (round 4, GPU call 20)" --out "$OUT/pmc_traffic.json" && cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json
rm -rf "$OUT/pmc1" "$OUT/tr1"
TAILN=4 run pytest_gpu 900 python -m pytest tests -q -m gpu
run smoke 200 python __graft_entry__.py --smoke
run bench 900 python bench.py --steps 20 --warmup 5
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
for f in "$OUT"/frame_*.log; do echo "$(basename $f): $(grep -h sampling $f | cut -c1-120)"; done
grep -h "cp_attn_o" "$OUT"/ts_front.log | cut -c1-250
cat "$OUT/summary.txt"
