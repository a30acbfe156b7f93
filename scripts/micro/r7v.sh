# after giving the reverb builds of a one-kernel run their own registers back (128 only in the variants a two-kernel step launches): headline against the all-120 build of r7u, configs[2], configs[2] tests
mkdir -p gpurun_out/r7v
for rep in 1 2 3; do for lib in oalsfxpp_amd/csrc/liboalsfx_hip.so ab/liboalsfx_hip_nep.so; do
OALSFX_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 400 --no-cpu-baseline --host-io 0 --no-other-configs --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib 400 steps', d['ms_per_step'], d['value'])"
OALSFX_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-io 0 --no-other-configs --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib driver-20', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r7v/headline_registers.txt
for rep in 1 2; do for flags in 0 0x400; do
OALSFX_DEBUG_FLAGS=$flags timeout -k 10 300 python bench.py --workload config3 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config3 flags $flags', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r7v/config3.txt
timeout -k 10 600 python -m pytest tests/test_gpu_chained.py -q -k "two_launches or configs_2" 2>&1 | tail -2
