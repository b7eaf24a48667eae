#!/usr/bin/env bash
# Round 3, GPU call 10: validation + measurement of the round's final build -- full GPU suite, the bench line as the driver runs it,
# rocprofv3 kernel trace of the bench command, PMC passes (FETCH_SIZE; SQ wave / wait counters; instruction-cache requests of the
# frame step), in-kernel timestamps, configs 2 / 4 / 5 and the 60 s utterance, codec kernel trace + MFMA-busy counters.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3j
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
TAILN=6 run pytest_gpu 1100 python -m pytest tests -q -m gpu -s
run bench 420 python bench.py --gpus 1 --steps 20 --warmup 5
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
run perf_frame 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker --prof
run perf_frame_06b 200 python tools/perf_frame.py --model 0.6b --frames 60 --talker
prof rocprof_bench --kernel-trace --stats -d "$PWD/$OUT/prof" -o perf -- python "$PWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-roofline
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof"
tail -14 "$OUT/kernel_trace.md"
prof pmc_fetch --kernel-trace --pmc FETCH_SIZE -d "$PWD/$OUT/pmc1" -o pmc -- python "$PWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph
DB=$(find "$OUT/pmc1" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_fetch_size.md" > /dev/null 2>&1; rm -rf "$OUT/pmc1"
prof pmc_sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d "$PWD/$OUT/pmc2" -o pmc -- python "$PWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph
DB=$(find "$OUT/pmc2" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_sq_frame_kernels.md" > /dev/null 2>&1; rm -rf "$OUT/pmc2"
prof pmc_icache --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d "$PWD/$OUT/pmc3" -o pmc -- python "$PWD/tools/perf_frame.py" --model 1.7b --frames 4 --talker --reps 1 --no-graph
DB=$(find "$OUT/pmc3" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_icache_frame_kernels.md" > /dev/null 2>&1; rm -rf "$OUT/pmc3"
head -16 "$OUT/pmc_icache_frame_kernels.md"
QTTS_LIBRARY=$PWD/qwen3-tts_amd/libqtts_tstamp.so TAILN=20 run ts_graph 200 python tools/ts_frame.py --model 1.7b --frames 12 --json "$OUT/ts_graph.json"
run codec_only 300 python tools/bench_configs.py codec_only --trials 10
run first_packet 240 python tools/bench_configs.py first_packet --trials 30
run clone_shard_e1 300 python tools/bench_configs.py clone_shard
run clone_shard_e2 300 python tools/bench_configs.py clone_shard --engines 2
run bench_clone 300 python bench.py --workload clone-shard --steps 1 --warmup 1
run long 400 python tools/bench_configs.py long --frames 750
prof rocprof_codec --kernel-trace --stats -d "$PWD/$OUT/prof2" -o perf -- python "$PWD/tools/perf_frame.py" --codec --reps 3 --batch 8
DB=$(find "$OUT/prof2" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/codec_kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof2"
head -10 "$OUT/codec_kernel_trace.md"
prof pmc_mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d "$PWD/$OUT/pmc4" -o pmc -- python "$PWD/tools/perf_frame.py" --codec --reps 1 --batch 8
DB=$(find "$OUT/pmc4" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" --out "$OUT/pmc_mfma_codec.md" > /dev/null 2>&1; rm -rf "$OUT/pmc4"
grep -E "resunit|gemm_tap2" "$OUT/pmc_mfma_codec.md" | head -12
cat "$OUT/summary.txt"
