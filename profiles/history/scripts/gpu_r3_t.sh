#!/usr/bin/env bash
# Round 3, GPU call 21: the new GPU test (bf16 hand-over in the prefill: bit-identical on hardware; call 20 ran it with the engine's
# default seeded sampling and compared three different draws), and the batch-32 frame step from inside: config 4's kernel trace
# (call 20, profiles/r03_config4_kernel_trace.md) shows its decode GEMMs at ~10 us per launch, twice the batch-8 kernel's.
# (Call 20 also ran smoke() -- ok -- and `rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py first_packet --trials 6`.)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3t
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -q -m gpu -x -k "prefill_bf16_handover or talker_bf16_mode_tracks" > "$OUT/pytest_new.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_new.log"
QTTS_LIBRARY=$PWD/qwen3-tts_amd/libqtts_tstamp.so timeout 200 python tools/ts_frame.py --model 1.7b --frames 6 --batch 32 > "$OUT/ts_frame_b32.log" 2>&1; echo "ts rc=$?"; grep -v amdgpu.ids "$OUT/ts_frame_b32.log" | cut -c1-170
timeout 200 python tools/perf_frame.py --model 1.7b --frames 30 --talker --batch 32 > "$OUT/perf_frame_b32.log" 2>&1; grep "ms/frame" "$OUT/perf_frame_b32.log" | cut -c1-120
