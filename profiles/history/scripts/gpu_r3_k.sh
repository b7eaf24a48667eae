#!/usr/bin/env bash
# Round 3, GPU call 11: (1) persistent launch + grid barriers vs a hipGraph chain of launches on the code predictor's GEMM chain
# (tools/persist_probe.py); (2) A/B of the codec's tap-reuse GEMM: weight tiles requested three steps ahead (the tree) vs two
# (libqtts_tap2old.so = the tree with the previous gemm_tap.hip), interleaved runs of config 2 / the 8 x 10 s decode.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3k
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
: > "$OUT/summary.txt"
TAILN=10 run persist_probe 120 python tools/persist_probe.py
for i in 1 2 3; do
  run codec_new_$i 120 python tools/bench_configs.py codec_only
  QTTS_LIBRARY=$PWD/qwen3-tts_amd/libqtts_tap2old.so run codec_old_$i 120 python tools/bench_configs.py codec_only
done
for f in "$OUT"/codec_*.log; do echo "$f: $(grep -o '"dtype": "bf16", "batch": [18], "ms_p50": [0-9.]*' "$f" | tr '\n' ' ')"; done | tee -a "$OUT/summary.txt"
