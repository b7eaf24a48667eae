#!/usr/bin/env bash
# Round 4, GPU call 17: cp_attn_o with the first read delayed (A/B against the two launches, timeline with the launch length), then -- the
# fused launch being opt-in -- the stamped counters / trace of the default tree again (talker_engine.hip changed) and the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4p
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
QTTS_CP_ATTN_O=1 run frame_fused_1 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
run frame_plain_1 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
QTTS_CP_ATTN_O=1 run frame_fused_2 150 python tools/perf_frame.py --model 1.7b --frames 60 --talker --reps 3
TAILN=30 QTTS_LIBRARY_OK=1 QTTS_CP_ATTN_O=1 run ts_fused 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_fused.json"
TAILN=6 run pytest_fused 600 python -m pytest tests -q -m gpu -x -s -k "fused_attention or tiny_greedy"
prof pmc_fetch --kernel-trace --pmc FETCH_SIZE -d "$PWD/$OUT/pmc1" -o pmc -- python "$PWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph
prof trace_bench --kernel-trace --stats -d "$PWD/$OUT/tr1" -o perf -- python "$PWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --no-api-e2e --no-roofline
FDB=$(find "$OUT/pmc1" -name "*.db" | head -1); TDB=$(find "$OUT/tr1" -name "*.db" | head -1)
[ -n "$FDB" ] && python tools/rocpd_pmc.py "$FDB" --out "$OUT/pmc_fetch_size.md" > /dev/null 2>&1
[ -n "$TDB" ] && python tools/rocpd_stats.py "$TDB" --out "$OUT/rocprofv3_kernel_trace_bench.md" > /dev/null 2>&1
[ -n "$FDB" ] && [ -n "$TDB" ] && python tools/pmc_traffic.py --fetch-db "$FDB" --trace-db "$TDB" --source "profiles/r04_pmc_fetch_size.md + profiles/r04_rocprofv3_kernel_trace_bench.md (round 4, GPU call 17)" --out "$OUT/pmc_traffic.json" && cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json
rm -rf "$OUT/pmc1" "$OUT/tr1"
run bench 900 python bench.py --steps 20 --warmup 5
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
grep -h "sampling\|greedy" "$OUT"/frame_*.log | cut -c1-170
grep -h "cp_attn_o" "$OUT"/pytest_fused.log "$OUT"/ts_fused.log | cut -c1-250
cat "$OUT/summary.txt"
