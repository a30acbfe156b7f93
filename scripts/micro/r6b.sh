# steps of two launches chained: the new tests, then config 3 chained against stream order
mkdir -p gpurun_out/r6b
timeout -k 10 900 python -m pytest tests/test_gpu_chained.py -x -q -k "two_launches" > gpurun_out/r6b/tests.log 2>&1; echo "tests exit $?" | tee -a gpurun_out/r6b/tests.log
tail -5 gpurun_out/r6b/tests.log
for rep in 1 2; do for flags in 0 0x400; do
OALSFX_DEBUG_FLAGS=$flags timeout -k 10 300 python bench.py --workload config3 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config3 flags $flags', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r6b/config3_chained.txt
