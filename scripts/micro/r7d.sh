# the reverb groups' wavefronts one priority level above the ring-light ones (ab/liboalsfx_hip_rbp1.so) against the product: the mixed grid chained at 4096 and 8192 instances, configs[2], the headline
mkdir -p gpurun_out/r7d
for rep in 1 2; do for lib in oalsfxpp_amd/csrc/liboalsfx_hip.so ab/liboalsfx_hip_rbp1.so; do
for n in 4096 8192; do
OALSFX_LIB=$PWD/$lib OALSFX_DEBUG_FLAGS=0x8000 timeout -k 10 300 python bench.py --workload config4 --instances $n --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib config4 x $n chained', d['ms_per_step'], d['value'])"
done
OALSFX_LIB=$PWD/$lib OALSFX_DEBUG_FLAGS=0x400 timeout -k 10 300 python bench.py --workload config4 --instances 4096 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib config4 x 4096 stream order', d['ms_per_step'], d['value'])"
OALSFX_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload config3 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib config3', d['ms_per_step'], d['value'])"
OALSFX_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --host-io 0 --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib headline', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r7d/reverb_base_priority.txt
