#!/usr/bin/env bash
# Round 4, GPU call 25: talker_engine.hip gained the two-engine cap for the fused launch (host logic) -- the stamped decode-GEMM passes again for
# this tree's digest, and the fused launch's GPU test.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4x
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
prof pmc_fetch --kernel-trace --pmc FETCH_SIZE -d "$PWD/$OUT/pmc1" -o pmc -- python "$PWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph
prof trace_bench --kernel-trace --stats -d "$PWD/$OUT/tr1" -o perf -- python "$PWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --no-api-e2e --no-roofline
FDB=$(find "$OUT/pmc1" -name "*.db" | head -1); TDB=$(find "$OUT/tr1" -name "*.db" | head -1)
[ -n "$FDB" ] && python tools/rocpd_pmc.py "$FDB" --out "$OUT/pmc_fetch_size.md" > /dev/null 2>&1
[ -n "$TDB" ] && python tools/rocpd_stats.py "$TDB" --out "$OUT/rocprofv3_kernel_trace_bench.md" > /dev/null 2>&1
[ -n "$FDB" ] && [ -n "$TDB" ] && python tools/pmc_traffic.py --fetch-db "$FDB" --trace-db "$TDB" --source "profiles/r04_pmc_fetch_size.md + profiles/r04_rocprofv3_kernel_trace_bench.md (round 4, GPU call 25)" --out "$OUT/pmc_traffic.json"
rm -rf "$OUT/pmc1" "$OUT/tr1"
timeout 120 python -m pytest tests -q -m gpu -x -k "fused_attention" 2>&1 | tail -2 | tee -a "$OUT/summary.txt"
cat "$OUT/summary.txt"
