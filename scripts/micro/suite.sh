# the whole GPU suite as the driver runs it, log under gpurun_out/$1
mkdir -p gpurun_out/$1
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/$1/gpu_suite.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/$1/gpu_suite.log; grep -E "^E  |^FAILED" gpurun_out/$1/gpu_suite.log | head -30
