mkdir -p gpurun_out/r4l
timeout -k 10 1100 python -m pytest tests/test_gpu_chained.py -q -m gpu > gpurun_out/r4l/chained.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r4l/chained.log; grep -E "^E  " gpurun_out/r4l/chained.log | head -20
