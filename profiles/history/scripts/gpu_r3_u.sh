#!/usr/bin/env bash
# Round 3, GPU call 22: batch 17..32 decode GEMMs through straight-line instantiations (skinny2_kernel<2, ..., NCH = 1..4>) vs the
# chunk loop (QTTS_SKINNY2_STRAIGHT_MT2=0): batch-32 frame step and first packet, alternating on one box; in-kernel timestamps; the
# GPU tests that run batches above 8.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3u
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
for i in 1 2; do
  timeout 200 python tools/perf_frame.py --model 1.7b --frames 30 --talker --batch 32 > "$OUT/frame_b32_new_$i.log" 2>&1
  QTTS_SKINNY2_STRAIGHT_MT2=0 timeout 200 python tools/perf_frame.py --model 1.7b --frames 30 --talker --batch 32 > "$OUT/frame_b32_old_$i.log" 2>&1
done
for f in "$OUT"/frame_b32_*.log; do echo "$f: $(grep -h 'ms/frame' "$f" | cut -c1-72 | tr '\n' ' ')"; done
timeout 200 python tools/bench_configs.py first_packet --trials 20 > "$OUT/first_packet_new.log" 2>&1; tail -1 "$OUT/first_packet_new.log" | cut -c1-260
QTTS_SKINNY2_STRAIGHT_MT2=0 timeout 200 python tools/bench_configs.py first_packet --trials 20 > "$OUT/first_packet_old.log" 2>&1; tail -1 "$OUT/first_packet_old.log" | cut -c1-260
QTTS_LIBRARY=$PWD/qwen3-tts_amd/libqtts_tstamp.so timeout 200 python tools/ts_frame.py --model 1.7b --frames 6 --batch 32 > "$OUT/ts_frame_b32.log" 2>&1; grep -v amdgpu.ids "$OUT/ts_frame_b32.log" | cut -c1-170
timeout 900 python -m pytest tests -q -m gpu -x -k "large_batch or batch32 or B_20 or bf16_mode_tracks or streaming_text" > "$OUT/pytest_batch.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_batch.log"
