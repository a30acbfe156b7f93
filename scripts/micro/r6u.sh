# the mixed grid (configs[3]) and the preset mix on two streams in turn against three
mkdir -p gpurun_out/r6u
for rep in 1 2; do for d in 2 3; do
OALSFX_CHAIN_DEPTH=$d timeout -k 10 300 python bench.py --workload config4 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config4 streams $d', d['ms_per_step'], d['value'])"
OALSFX_CHAIN_DEPTH=$d timeout -k 10 300 python bench.py --preset-mix --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('preset mix streams $d', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r6u/depth.txt
