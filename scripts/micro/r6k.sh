# steps of two launches (BASELINE configs[2]: chorus -> flanger -> echo -> EAX reverb) chained against stream order, by batch size
mkdir -p gpurun_out/r6k
for n in 1024 2048 3072 4096 6144 8192; do for flags in 0 0x400; do
OALSFX_DEBUG_FLAGS=$flags timeout -k 10 300 python bench.py --workload config3 --instances $n --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config3 x $n flags $flags', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r6k/config3_by_size.txt
