# ring-light batches chained against stream order (4096 instances of one type, 256-frame calls), after the tests
mkdir -p gpurun_out/r6i
timeout -k 10 900 python -m pytest tests/test_gpu_chained.py -x -q -k "eleven_types or configs_3_chains or ring_light_effects or two_launches" > gpurun_out/r6i/tests.log 2>&1; echo "tests exit $?" | tee -a gpurun_out/r6i/tests.log
tail -5 gpurun_out/r6i/tests.log
timeout -k 10 600 python scripts/light_chain_bench.py | tee gpurun_out/r6i/ring_light_chained.txt
