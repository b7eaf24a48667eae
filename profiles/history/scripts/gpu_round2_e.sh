#!/usr/bin/env bash
# Round 2, GPU call 7: validation of the round's final default build -- full GPU suite, bench line, rocprofv3 kernel trace of the
# bench command, FETCH_SIZE pass of the decode GEMM, long utterance with split-KV attention, codec config 2, MFMA-busy counters.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2e
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n 3 "$OUT/$name.log" | cut -c1-400 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
run pytest_gpu 900 python -m pytest tests -q -m gpu -s
run bench 420 python bench.py --steps 5 --warmup 2
run long 420 python tools/bench_configs.py long --frames 750
run long_nosplit 420 env QTTS_ATTN_NSPLIT=1 python tools/bench_configs.py long --frames 750
run codec_only 300 python tools/bench_configs.py codec_only --trials 10
run ab 300 python tools/ab_inproc.py --frames 40 --reps 2
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o perf -- python "$OLDPWD/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/$OUT/pmc1" -o pmc -- python "$OLDPWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph > "$OLDPWD/$OUT/pmc_fetch.log" 2>&1 ); echo "pmc_fetch rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/pmc1" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_fetch_size.md" > /dev/null 2>&1; rm -rf "$OUT/pmc1"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d "$OLDPWD/$OUT/pmc2" -o pmc -- python "$OLDPWD/tools/perf_frame.py" --codec --reps 1 --batch 8 > "$OLDPWD/$OUT/pmc_mfma.log" 2>&1 ); echo "pmc_mfma rc=$?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/pmc2" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_mfma_codec.md" > /dev/null 2>&1; rm -rf "$OUT/pmc2"
cat "$OUT/summary.txt"
