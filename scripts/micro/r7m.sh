mkdir -p gpurun_out/r7m
OALSFX_CHAIN_FUZZ_FIRST=5000 OALSFX_CHAIN_FUZZ_SEEDS=${1:-800} timeout -k 10 1100 python -m pytest tests/test_gpu_chained.py -x -q -k "test_random_runs and not other_shapes" > gpurun_out/r7m/fuzz_multichannel.log 2>&1; echo "exit $?" >> gpurun_out/r7m/fuzz_multichannel.log
tail -12 gpurun_out/r7m/fuzz_multichannel.log
