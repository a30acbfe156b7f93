# chained steps of two launches: order of the ring-light launch (0x100 at the time, 0x10 since: its own list), declared workgroup sizes (0x1000), against stream order (0x400)
mkdir -p gpurun_out/r6f
for rep in 1 2 3; do for flags in 0 0x100 0x1000 0x400; do
OALSFX_DEBUG_FLAGS=$flags timeout -k 10 300 python bench.py --workload config3 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config3 flags $flags', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r6f/config3_variants.txt
