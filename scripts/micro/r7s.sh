# fused runs of reverb-free slots: the slots' effect types asked for together (product) against a load in front of each slot (ab/liboalsfx_hip_ta0.so): configs[2], alternating
mkdir -p gpurun_out/r7s
for rep in 1 2 3; do for lib in oalsfxpp_amd/csrc/liboalsfx_hip.so ab/liboalsfx_hip_ta0.so; do
OALSFX_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload config3 --steps 300 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib config3', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r7s/types_ahead.txt
