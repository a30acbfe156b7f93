mkdir -p gpurun_out/r4s
timeout -k 10 300 python3 scripts/host_pipeline_forms.py 2>&1 | tee gpurun_out/r4s/host_pipeline_forms.txt
timeout -k 10 600 python -m pytest tests/test_gpu_async_and_ranks.py -q -m gpu 2>&1 | tail -4
OALSFX_HOST_PIPELINE=1 timeout -k 10 600 python -m pytest tests/test_gpu_async_and_ranks.py -q -m gpu -k "async or pipelin" 2>&1 | tail -3
timeout -k 10 300 python bench.py --in-process --devices 0,0 --steps 100 --warmup 20 2>&1 | tail -1 | tee gpurun_out/r4s/bench_in_process_two_shards.json
timeout -k 10 300 python bench.py --in-process --steps 100 --warmup 20 2>&1 | tail -1 | tee gpurun_out/r4s/bench_in_process_one_shard.json
