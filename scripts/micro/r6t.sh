# two-launch steps and the mixed grid on three streams in turn (0x10000) against two
mkdir -p gpurun_out/r6t
for rep in 1 2; do for w in config3 config4; do for flags in 0 0x10000; do
OALSFX_DEBUG_FLAGS=$flags timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w flags $flags', d['ms_per_step'], d['value'])"
done; done; done | tee gpurun_out/r6t/depth.txt
