set -x
mkdir -p gpurun_out/c15
timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "one_utterance_greedy or 17b_ragged_batch or batch8_10s_greedy or voice_clone_icl or tiny_greedy_bit_exact or vs_oracle_fresh" > gpurun_out/c15/pytest_f32.log 2>&1; echo "pytest rc $?" >> gpurun_out/c15/pytest_f32.log
tail -5 gpurun_out/c15/pytest_f32.log
timeout 300 python tools/ab_inproc.py --dtype f32 --frames 40 --reps 3 --only default cp_mlp_off > gpurun_out/c15/ab_f32.log 2>&1
tail -8 gpurun_out/c15/ab_f32.log
