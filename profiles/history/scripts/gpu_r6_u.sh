#!/usr/bin/env bash
# Round 6, call u: where config 4's first packet and config 2's decode spend their time on this tree: kernel traces (per-kernel tables).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6u
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
: > "$OUT/summary.txt"
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
prof trace_c4 --kernel-trace --stats -d "$PWD/$OUT/tr1" -o perf -- python "$PWD/tools/bench_configs.py" first_packet --trials 6
DB=$(find "$OUT/tr1" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/config4_kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/tr1"
prof trace_b1 --kernel-trace --stats -d "$PWD/$OUT/tr2" -o perf -- python "$PWD/tools/perf_frame.py" --codec --reps 5 --batch 1
DB=$(find "$OUT/tr2" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/codec_b1_kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/tr2"
tail -3 "$OUT/trace_c4.log" | cut -c1-600
for B in 1 8; do timeout 300 python tools/perf_frame.py --codec --reps 5 --batch $B 2>&1 | grep "codec bf16" | tee -a "$OUT/summary.txt"; done
