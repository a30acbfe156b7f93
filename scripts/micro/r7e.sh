# how often a waiting wavefront looks at its turn word: s_sleep 2 (product) against 0 / 1 / 6, the headline chained, 400 steps, alternating
mkdir -p gpurun_out/r7e
for rep in 1 2 3; do for lib in oalsfxpp_amd/csrc/liboalsfx_hip.so ab/liboalsfx_hip_ts0.so ab/liboalsfx_hip_ts1.so ab/liboalsfx_hip_ts6.so; do
OALSFX_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 400 --no-cpu-baseline --host-io 0 --no-other-configs --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib headline 400 steps', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r7e/turn_sleep.txt
