#!/usr/bin/env bash
# Round 6, call l: the prefill's attention with four query rows per wave (attn_rows4_kernel): the fp32 goldens (bit-exact codes through it), config 4
# (first packet: 12 of its 31 ms are the batch-32 prefill) with and without, kernel trace of the prefill.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6l
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
TAILN=8 run pytest_golden 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "vs_reference_golden and not fp32_instantiations"
TAILN=2 run config4 600 python tools/bench_configs.py first_packet
TAILN=2 run config4_off 600 env QTTS_ATTN_ROWS4=0 python tools/bench_configs.py first_packet
TAILN=2 run config4_b 600 python tools/bench_configs.py first_packet
TAILN=2 run config4_off_b 600 env QTTS_ATTN_ROWS4=0 python tools/bench_configs.py first_packet
prof trace_c4 --kernel-trace --stats -d "$PWD/$OUT/tr1" -o perf -- python "$PWD/tools/bench_configs.py" first_packet --trials 4
DB=$(find "$OUT/tr1" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace_config4.md" > /dev/null 2>&1; rm -rf "$OUT/tr1"
grep -E "attn_rows|gemm_wide" "$OUT/kernel_trace_config4.md" | cut -c1-160
cat "$OUT/summary.txt"
