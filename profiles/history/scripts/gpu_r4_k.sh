#!/usr/bin/env bash
# Round 4, GPU call 12 (the round's last tree): codec graphs keyed on the shape with engine-owned staging -- the whole GPU suite, smoke(),
# first packet (p99: no capture inside a request any more), config 2, the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4k
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-700 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=4 run pytest_gpu 900 python -m pytest tests -q -m gpu
run smoke 200 python __graft_entry__.py --smoke
run first_packet 200 python tools/bench_configs.py first_packet --trials 30
QTTS_CODEC_GRAPH=0 run first_packet_nograph 200 python tools/bench_configs.py first_packet --trials 30
run codec_only 200 python tools/bench_configs.py codec_only --trials 10
run bench 900 python bench.py --steps 20 --warmup 5
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
cat "$OUT/summary.txt"
