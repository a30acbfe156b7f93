mkdir -p gpurun_out/r7a
OALSFX_CHAIN_FUZZ_FIRST=2000 OALSFX_CHAIN_FUZZ_SEEDS=${1:-800} timeout -k 10 1100 python -m pytest tests/test_gpu_chained.py -x -q -k "other_shapes" > gpurun_out/r7a/fuzz_other_shapes_rates.log 2>&1; echo "exit $?" >> gpurun_out/r7a/fuzz_other_shapes_rates.log
tail -25 gpurun_out/r7a/fuzz_other_shapes_rates.log
