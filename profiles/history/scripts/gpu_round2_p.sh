#!/usr/bin/env bash
# Round 2, GPU call 36 (last): bench step with codec packets overlapped with the AR loop (experiment) vs the reported configuration.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r2p
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$OUT/bench_base.log" 2>&1; echo "base rc=$?"
timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --overlap-codec 25 > "$OUT/bench_overlap25.log" 2>&1; echo "overlap rc=$?"
for f in bench_base bench_overlap25; do grep -h '^{' "$OUT/$f.log" | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$f', j['value'], j['ms_per_step'], j.get('ar_ms_per_frame'), j.get('codec_ms_per_step'))"; done
tail -3 "$OUT/bench_overlap25.log" | cut -c1-300
