# the mixed grid in ascending order of workgroup length (shortest type first, reverb groups last), chained (0x8000: whatever its size) and in stream order
mkdir -p gpurun_out/r6p
for n in 4096 8192; do for order in list ascending; do for flags in 0x8000 0x400; do
OALSFX_MIXED_ORDER=$order OALSFX_DEBUG_FLAGS=$flags timeout -k 10 300 python bench.py --workload config4 --instances $n --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config4 x $n order $order flags $flags', d['ms_per_step'], d['value'])"
done; done; done | tee gpurun_out/r6p/config4_grid_order.txt
