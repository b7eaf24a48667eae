#!/usr/bin/env bash
# Round 4, GPU call 5: gemm_dma_kernel (both operands by LDS-DMA, one barrier per step) -- correctness on hardware (bf16 codec and prefill
# tests with the kernel forced), microbenchmark against gemm_wide on the prefill's shapes, codec and first packet with / without.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r4e
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-1200 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
QTTS_GEMM_DMA=1 TAILN=6 run pytest_dma 600 python -m pytest tests -q -m gpu -x -s -k "bf16 or codec_real or prefill"
TAILN=12 run gemm_ab 300 python tools/bench_gemm_dma.py
run codec_off 200 python tools/bench_configs.py codec_only --trials 10
QTTS_GEMM_DMA=1 run codec_dma 200 python tools/bench_configs.py codec_only --trials 10
run fp_off 200 python tools/bench_configs.py first_packet --trials 10
QTTS_GEMM_DMA=2 run fp_dma_prefill 200 python tools/bench_configs.py first_packet --trials 10
QTTS_GEMM_DMA=1 run fp_dma_all 200 python tools/bench_configs.py first_packet --trials 10
cat "$OUT/summary.txt"
