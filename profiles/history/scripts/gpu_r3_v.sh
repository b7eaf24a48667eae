#!/usr/bin/env bash
# Round 3, GPU call 23: validation of the round's LAST tree (call 19's build + the batch 17..32 straight-line decode GEMMs + the new GPU
# test): full GPU suite, smoke(), the bench line as the driver runs it, rocprofv3 kernel trace of the bench command, configs 2 and 4.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3v
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_gpu 1100 python -m pytest tests -q -m gpu -s
run smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
run bench 420 python bench.py --gpus 1 --steps 20 --warmup 5
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
prof rocprof_bench --kernel-trace --stats -d "$PWD/$OUT/prof" -o perf -- python "$PWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-roofline
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof"
tail -3 "$OUT/kernel_trace.md"
run codec_only 300 python tools/bench_configs.py codec_only --trials 10
run first_packet 240 python tools/bench_configs.py first_packet --trials 30
cat "$OUT/summary.txt"
