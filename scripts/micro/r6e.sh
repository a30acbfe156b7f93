# workgroups of one size for the two kernels of a chained step: timeline, tests, config 3 chained against stream order, and the headline (whose kernel now takes 128 registers)
mkdir -p gpurun_out/r6e
timeout -k 10 400 bash scripts/chain_trace.sh r6e 4096 256 40 config3 > /dev/null 2>&1; tail -22 gpurun_out/r6e/timeline.txt; grep step gpurun_out/r6e/probe.log
timeout -k 10 900 python -m pytest tests/test_gpu_chained.py -x -q -k "two_launches" > gpurun_out/r6e/tests.log 2>&1; echo "tests exit $?" | tee -a gpurun_out/r6e/tests.log
tail -3 gpurun_out/r6e/tests.log
for rep in 1 2; do for flags in 0 0x400; do
OALSFX_DEBUG_FLAGS=$flags timeout -k 10 300 python bench.py --workload config3 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config3 flags $flags', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r6e/config3_chained.txt
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('headline 200', d['ms_per_step'], d['value'], d['roofline']['achieved'])"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('headline driver-20', d['ms_per_step'], d['value'])"
done | tee gpurun_out/r6e/headline.txt
