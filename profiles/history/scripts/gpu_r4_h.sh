#!/usr/bin/env bash
# Round 4, GPU call 8: attn_rows with 16 keys per wave and loop trip (prefill / codec transformer attention) -- goldens, first packet, codec;
# strip-width floor of the decode GEMM at batch 32 (QTTS_FS_MIN_WGS = 48 / 96 / 192).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4h
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-1200 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=5 run pytest_attn 900 python -m pytest tests -q -m gpu -x -k "tiny_greedy or 06b_one or 17b_ragged or codec or prompt_assembly or prefill or base_voice_clone"
run fp 200 python tools/bench_configs.py first_packet --trials 10
run codec 200 python tools/bench_configs.py codec_only --trials 10
run b32_fs96 200 python tools/perf_frame.py --model 1.7b --frames 40 --talker --batch 32 --reps 3
QTTS_FS_MIN_WGS=48 run b32_fs48 200 python tools/perf_frame.py --model 1.7b --frames 40 --talker --batch 32 --reps 3
QTTS_FS_MIN_WGS=192 run b32_fs192 200 python tools/perf_frame.py --model 1.7b --frames 40 --talker --batch 32 --reps 3
run bench 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api-e2e --no-parity-mode
grep -h '^{' "$OUT/bench.log" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','ar_ms_per_frame','codec_ms_per_step')})"
grep -h "greedy\|sampling" "$OUT"/b32_*.log | cut -c1-170
cat "$OUT/summary.txt"
