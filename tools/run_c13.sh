mkdir -p gpurun_out/c13
python tools/diag_att_trace.py > gpurun_out/c13/trace.log 2>&1
grep -a "WAW\|stage 1 first" gpurun_out/c13/trace.log | cut -c1-420 | head -14
QTTS_LIBRARY=$PWD/qwen3-tts_amd/libqtts_nopk.so python tools/diag_contention4.py > gpurun_out/c13/diag4_nopk.log 2>&1
grep -a "diag4" gpurun_out/c13/diag4_nopk.log | cut -c1-300 | awk '!seen[$0]++'
python tools/diag_contention4.py > gpurun_out/c13/diag4_product.log 2>&1
grep -a "diag4" gpurun_out/c13/diag4_product.log | cut -c1-300 | awk '!seen[$0]++' | head -3
