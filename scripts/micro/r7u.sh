# the headline kernel at 128 registers (product: places of one size) against 120 (ab/liboalsfx_hip_nep.so: the gate in front of a chained launch then finds registers free on a full chip): 400 steps and the driver's 20, alternating
mkdir -p gpurun_out/r7u
for rep in 1 2 3 4; do for lib in oalsfxpp_amd/csrc/liboalsfx_hip.so ab/liboalsfx_hip_nep.so; do
OALSFX_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 400 --no-cpu-baseline --host-io 0 --no-other-configs --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib 400 steps', d['ms_per_step'], d['value'])"
OALSFX_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-io 0 --no-other-configs --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib driver-20', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r7u/registers_120_vs_128.txt
