# the mixed grid (BASELINE configs[3]: type 1 + i % 11, random properties) chained against stream order, by batch size
mkdir -p gpurun_out/r6j
for n in 2048 4096 6144 8192 16384; do for flags in 0 0x400; do
OALSFX_DEBUG_FLAGS=$flags timeout -k 10 300 python bench.py --workload config4 --instances $n --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config4 x $n flags $flags', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r6j/config4_by_size.txt
