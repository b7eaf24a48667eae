#!/usr/bin/env bash
# Round 2, GPU call 8: bytes-aware strip-width floor + batched sampler gathers; SQ counter pass over the frame step's kernels.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2f
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n 3 "$OUT/$name.log" | cut -c1-400 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run pytest_gpu 900 python -m pytest tests -q -m gpu -s
run perf_frame 240 python tools/perf_frame.py --model 1.7b --frames 60 --talker --prof
run bench 420 python bench.py --steps 5 --warmup 2
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o perf -- python "$OLDPWD/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d "$OLDPWD/$OUT/pmc" -o pmc -- python "$OLDPWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph > "$OLDPWD/$OUT/pmc_sq.log" 2>&1 ); echo "pmc_sq rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/pmc" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_sq.md" > /dev/null 2>&1; rm -rf "$OUT/pmc"
cat "$OUT/summary.txt"
