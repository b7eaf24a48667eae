#!/usr/bin/env bash
# Round 6, call z6: qknorm_rope_store with the rotation computed once per row: the talker's golden tests, config 4's first packet and its kernel table.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}"
OUT=gpurun_out/r6z6
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
: > "$OUT/summary.txt"
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "talker or prefill or prompt or wrapper or teacher or pinned" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)" | tee -a "$OUT/summary.txt"
timeout 600 python tools/bench_configs.py first_packet --trials 20 2>&1 | tail -1 | cut -c1-500 | tee -a "$OUT/summary.txt"
prof trace_c4 --kernel-trace --stats -d "$PWD/$OUT/tr1" -o perf -- python "$PWD/tools/bench_configs.py" first_packet --trials 6
DB=$(find "$OUT/tr1" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/config4_kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/tr1"
grep -E "qknorm|attn_rows4|rmsnorm_kernel<true>|gemm_ring_kernel<4, 1" "$OUT/config4_kernel_trace.md" | cut -c1-140 | tee -a "$OUT/summary.txt"
