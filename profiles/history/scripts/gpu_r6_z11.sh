#!/usr/bin/env bash
# Round 6, call z11: final validation of the last tree (32-row tiles of the wide-K GEMM for launches of <= 128 rows; checker-side thread fix): whole GPU suite, smoke, bench line with
# the rocprofv3 trace + FETCH_SIZE pass (profiles/pmc_traffic.json), config 4 trace, clone-shard job.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6z11
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
TAILN=4 run pytest_gpu 1700 python -m pytest tests -q -m gpu -s --durations=15
run smoke 200 python __graft_entry__.py --smoke
prof pmc_fetch --kernel-trace --pmc FETCH_SIZE -d "$PWD/$OUT/pmc1" -o pmc -- python "$PWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph
prof trace_bench --kernel-trace --stats -d "$PWD/$OUT/tr1" -o perf -- python "$PWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --no-api-e2e --no-roofline --no-configs
FDB=$(find "$OUT/pmc1" -name "*.db" | head -1); TDB=$(find "$OUT/tr1" -name "*.db" | head -1)
[ -n "$FDB" ] && python tools/rocpd_pmc.py "$FDB" --out "$OUT/pmc_fetch_size.md" > /dev/null 2>&1
[ -n "$TDB" ] && python tools/rocpd_stats.py "$TDB" --out "$OUT/rocprofv3_kernel_trace_bench.md" > /dev/null 2>&1
[ -n "$FDB" ] && [ -n "$TDB" ] && python tools/pmc_traffic.py --fetch-db "$FDB" --trace-db "$TDB" --source "profiles/r06_pmc_fetch_size.md + profiles/r06_rocprofv3_kernel_trace_bench.md (round 6, final GPU call)" --out "$OUT/pmc_traffic.json" && cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json
rm -rf "$OUT/pmc1" "$OUT/tr1"
run bench 900 python bench.py --steps 20 --warmup 5
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
prof trace_c4 --kernel-trace --stats -d "$PWD/$OUT/tr3" -o perf -- python "$PWD/tools/bench_configs.py" first_packet --trials 6
DB=$(find "$OUT/tr3" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/config4_kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/tr3"
timeout 600 python tools/bench_configs.py first_packet --trials 20 2>&1 | tail -1 | cut -c1-500 | tee "$OUT/config4_first_packet.json"
run clone 600 python bench.py --workload clone-shard --steps 1 --warmup 1
grep -h '^{' "$OUT/clone.log" > "$OUT/clone.json"
cat "$OUT/summary.txt"
