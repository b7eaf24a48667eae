#!/usr/bin/env bash
# Round 3, GPU call 1: full GPU suite on the round's groundwork build (new goldens: 1.7B Base ICL b8, b32 past the trailing text,
# codec bf16 yardstick; split-KV by live-length buckets; general top-p), the bench line with the per-launch roofline leg and the
# fp32 parity-mode leg, A/B of the kernarg-preload variant, configs 4 and 5 re-measured, rocprofv3 kernel trace of the bench command.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3a
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_gpu 1100 python -m pytest tests -q -m gpu -s
run bench 420 python bench.py --steps 5 --warmup 2
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
TAILN=12 run ab_kpre 500 python tools/ab_variants.py --frames 60 --only default kpre default_again
cp gpurun_out/ab/ab.json "$OUT/ab_kpre.json" 2>/dev/null
run first_packet 240 python tools/bench_configs.py first_packet --trials 30
run clone_shard_e1 300 python tools/bench_configs.py clone_shard
run clone_shard_e2 300 python tools/bench_configs.py clone_shard --engines 2
run bench_clone 300 python bench.py --workload clone-shard --steps 1 --warmup 1
( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o perf -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
grep -h '^{' "$OUT/rocprof.log" > "$OUT/bench_under_rocprof.json"
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof"
tail -12 "$OUT/kernel_trace.md"
cat "$OUT/summary.txt"
