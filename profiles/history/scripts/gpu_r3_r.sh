#!/usr/bin/env bash
# Round 3, GPU call 18: EXPERIMENT -- the talker's q|k|v GEMM touching the live K / V pages of the attention launch that follows
# (QTTS_KV_PREFETCH=1, skinny.hip / SkinnyParams::pf_k): alternating runs of the 1.7B frame step on one box.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3r
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
for i in 1 2 3; do
  QTTS_KV_PREFETCH=0 timeout 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker > "$OUT/frame_off_$i.log" 2>&1
  QTTS_KV_PREFETCH=1 timeout 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker > "$OUT/frame_on_$i.log" 2>&1
done
for f in "$OUT"/frame_*.log; do echo "$f: $(grep -h 'ms/frame' "$f" | cut -c1-70 | tr '\n' ' ')"; done
