# how much LDS the workgroups of a two-kernel run should ask for (config 3: ring-light 22 272 B declared, reverb 30 400)
mkdir -p gpurun_out/r6m
for rep in 1 2; do for lds in 40960 40448 38400 35840 33280 30720; do
OALSFX_EQUAL_LDS=$lds timeout -k 10 300 python bench.py --workload config3 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config3 lds $lds', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r6m/config3_lds.txt
