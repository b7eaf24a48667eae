#!/usr/bin/env bash
# Round 5, after comment-only edits of digest-covered sources (device code identical: the disassembly of all 325 kernels hashes the same):
# the stamped passes re-taken for the last tree's digest + the bench line + smoke.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r5s
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
run smoke 200 python __graft_entry__.py --smoke
prof pmc_fetch --kernel-trace --pmc FETCH_SIZE -d "$PWD/$OUT/pmc1" -o pmc -- python "$PWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph
prof trace_bench --kernel-trace --stats -d "$PWD/$OUT/tr1" -o perf -- python "$PWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --no-api-e2e --no-roofline --no-configs
FDB=$(find "$OUT/pmc1" -name "*.db" | head -1); TDB=$(find "$OUT/tr1" -name "*.db" | head -1)
[ -n "$FDB" ] && python tools/rocpd_pmc.py "$FDB" --out "$OUT/pmc_fetch_size.md" > /dev/null 2>&1
[ -n "$TDB" ] && python tools/rocpd_stats.py "$TDB" --out "$OUT/rocprofv3_kernel_trace_bench.md" > /dev/null 2>&1
[ -n "$FDB" ] && [ -n "$TDB" ] && python tools/pmc_traffic.py --fetch-db "$FDB" --trace-db "$TDB" --source "profiles/r05_pmc_fetch_size.md + profiles/r05_rocprofv3_kernel_trace_bench.md (round 5, last GPU call)" --out "$OUT/pmc_traffic.json" && cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json
rm -rf "$OUT/pmc1" "$OUT/tr1"
run bench 900 python bench.py --steps 20 --warmup 5
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
TAILN=2 run pytest_fused 400 python -m pytest tests -q -m gpu -k "fused or contention or pinned_at_the_metric or tiny_greedy"
cat "$OUT/summary.txt"
