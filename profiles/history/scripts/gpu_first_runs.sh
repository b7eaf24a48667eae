#!/usr/bin/env bash
# One gpurun call that gives the widened engines their FIRST hardware run, each in its own process and under its own
# timeout (a faulting kernel aborts only that process; nothing here can hang the box past its limit), then refreshes the
# bench line.  Everything lands in gpurun_out/first_runs/ and comes back with the call.
#
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_first_runs.sh'
#
# Order = cheapest evidence first: validated suite (must stay green) -> experimental tests one by one -> bench.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/first_runs
mkdir -p "$OUT"
export QTTS_EXPERIMENTAL=1 PYTHONUNBUFFERED=1
run() {   # run <name> <timeout-s> <command...>
    local name=$1 lim=$2; shift 2
    local t0=$(date +%s)
    timeout --signal=TERM --kill-after=10 "$lim" "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"
    tail -n 3 "$OUT/$name.log" | sed "s/^/    /"
}
: > "$OUT/summary.txt"
run validated_suite 600 env -u QTTS_EXPERIMENTAL python -m pytest tests -q -m gpu -x
for t in test_codec_incremental_stream_equals_forward test_codec_encoder_codes_vs_reference_golden \
         test_speaker_embedding_vs_oracle "test_talker_generate_stream_equals_generate" \
         test_wrapper_voice_clone_from_waveform_end_to_end test_wrapper_stream_custom_voice_equals_one_shot; do
    run "exp_$t" 180 python -m pytest tests/test_gpu_parity.py -q -x -s -k "$t"
done
run bench 420 python bench.py --steps 3 --warmup 1
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json" || true
cat "$OUT/summary.txt"
