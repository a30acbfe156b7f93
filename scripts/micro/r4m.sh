mkdir -p gpurun_out/r4m
timeout -k 10 240 python -m pytest tests/test_gpu_chained.py -q -m gpu -k "same_cu or before_the_turn or every_instance_through" > gpurun_out/r4m/new_chained.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r4m/new_chained.log; grep -E "^E  " gpurun_out/r4m/new_chained.log | head -20
