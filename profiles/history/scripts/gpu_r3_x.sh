#!/usr/bin/env bash
# Round 3, GPU call 25: SwiGLU in ONE strip per workgroup at batch <= 8 (ACT_SWIGLU8: 768 instead of 384 workgroups for the talker's
# gate|up; microbenchmark of call 24: 9.5 vs 10.7 us streamed).  A/B of the frame step (QTTS_SWIGLU8 is read when the engine is
# built: alternating processes on one box), then the validation of this tree: full GPU suite, smoke(), bench line, rocprofv3 kernel
# trace of the bench command.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r3x
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
run() { local name=$1 lim=$2; shift 2; local t0=$(date +%s)
        timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
        echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"; tail -n ${TAILN:-3} "$OUT/$name.log" | cut -c1-600 | sed "s/^/    /"; }
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 "$@" > "$OLDPWD/$OUT/$name.log" 2>&1 ); echo "$name rc=$?" | tee -a "$OUT/summary.txt"; }
: > "$OUT/summary.txt"
for i in 1 2 3; do
  QTTS_SWIGLU8=1 timeout 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker > "$OUT/frame_on_$i.log" 2>&1
  QTTS_SWIGLU8=0 timeout 200 python tools/perf_frame.py --model 1.7b --frames 60 --talker > "$OUT/frame_off_$i.log" 2>&1
done
for f in "$OUT"/frame_o*.log; do echo "$f: $(grep -h 'ms/frame' "$f" | cut -c1-70 | tr '\n' ' ')"; done | tee -a "$OUT/summary.txt"
TAILN=6 run pytest_gpu 1100 python -m pytest tests -q -m gpu -s
run smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
run bench 420 python bench.py --gpus 1 --steps 20 --warmup 5
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json"
prof rocprof_bench --kernel-trace --stats -d "$PWD/$OUT/prof" -o perf -- python "$PWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-roofline
DB=$(find "$OUT/prof" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" --out "$OUT/kernel_trace.md" > /dev/null 2>&1; rm -rf "$OUT/prof"
tail -13 "$OUT/kernel_trace.md" | cut -c1-110
cat "$OUT/summary.txt"
