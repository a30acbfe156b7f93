# single grids of ring-light effects (alone or beside proven reverbs) as chained launches: tests, then config 4 and ring-light batches chained against stream order
mkdir -p gpurun_out/r6g
timeout -k 10 900 python -m pytest tests/test_gpu_chained.py -x -q -k "eleven_types or configs_3_chains or ring_light_effects_alone or two_launches" > gpurun_out/r6g/tests.log 2>&1; echo "tests exit $?" | tee -a gpurun_out/r6g/tests.log
tail -30 gpurun_out/r6g/tests.log
for rep in 1 2; do for flags in 0 0x400; do
OALSFX_DEBUG_FLAGS=$flags timeout -k 10 300 python bench.py --workload config4 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config4 flags $flags', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r6g/config4_chained.txt
