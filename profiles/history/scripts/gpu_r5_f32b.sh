set -x
mkdir -p gpurun_out/c16
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "fp32_instantiations" > gpurun_out/c16/pytest_f32.log 2>&1; echo "pytest rc $?" >> gpurun_out/c16/pytest_f32.log
tail -8 gpurun_out/c16/pytest_f32.log
timeout 300 python tools/ab_inproc.py --dtype f32 --frames 40 --reps 3 --only default f32_fused_mlp f32_fused_both > gpurun_out/c16/ab_f32.log 2>&1
tail -11 gpurun_out/c16/ab_f32.log
