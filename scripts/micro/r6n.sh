mkdir -p gpurun_out/r6n
OALSFX_CHAIN_FUZZ_SEEDS=${1:-300} timeout -k 10 1100 python -m pytest tests/test_gpu_chained.py -x -q -k "other_shapes" > gpurun_out/r6n/fuzz_other_shapes.log 2>&1; echo "exit $?" >> gpurun_out/r6n/fuzz_other_shapes.log
tail -25 gpurun_out/r6n/fuzz_other_shapes.log
